// veloxseg_amd._vxops -- the host-side operator bodies in C++ (pybind11 + ATen for allocation only).
//
// Why: one eager training step makes ~300 operator calls and ~1050 kernel launches; with the operator bodies in Python the host needs
// 14-16 ms per step, more than the GPU.  This module runs the SAME sequences of C-ABI calls (include/veloxseg_hip.h) that
// veloxseg_amd/functional.py runs -- routing, workspace allocation, launches -- without the interpreter in between.  Autograd stays in
// Python: every entry here is a plain forward or backward function returning tensors plus an opaque state object; functional.py keeps one
// torch.autograd.Function per operator / composite block and calls these from its forward / backward.  The arithmetic lives in
// libveloxseg_hip.so either way; this file contains no kernels and no CPU implementation.
//
// Reference call sites mirrored by the composites: JLC block conv_blocks.py:41-75; FFN tail PWA.py:437 + attention_utils.py:45-71.
#include <torch/extension.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/custom_class.h>
#include <torch/library.h>
#include <c10/hip/HIPStream.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <map>
#include <unordered_map>
#include <tuple>
#include <functional>
#include <type_traits>
#include <memory>
#include <vector>
#include <map>
#include <tuple>
#include "../../include/veloxseg_hip.h"

using at::Tensor;
typedef c10::optional<Tensor> OptT;

namespace {

struct Flags {
    bool fuse_blocks = true, use_s1 = true, use_conv_mfma = true, bf16_expand = false, use_expand_mfma = true, use_gconv1 = true, use_wgrad_ws = false, use_patchify = true, use_in_row = true, fuse_gelu = true, fuse_pw_bwd = true, use_down_mfma = true, skip_in_bias = true, in_split2 = true, fuse_res = true, fuse_bwd_add = true;
    int64_t pw_mfma_max_v = 4096, in_row_max = 4096;
    int tile_min_c = 32;              // JLC channel stage on the tile-GEMM kernels (pwa_fused.hip vx_inmlp_*) from this many channels up, mlp.hip below (functional.TILE_MIN_C)
    bool upconv_wgrad_mfma = true;    // (A/B) ConvTranspose weight gradient as one MFMA GEMM (pointwise.hip vx_upconv_k2s2_wgrad) vs the generic strided-conv kernel
    bool jlc_tz = true;               // JLC grouped convolutions (forward + input gradient) as Toeplitz GEMMs on the bf16 matrix pipe, fp32-exact products (csrc/jlc_mfma.hip); 0 = the fp32 VALU kernels of jlc.hip
    bool jlc_wg_tz = true;            // the three JLC weight gradients in one matrix-pipe launch (csrc/jlc_mfma.hip vx_jlc_wgrad_tz); 0 = the VALU kernels of conv_wgrad.hip
    bool jlc_tile = true;             // JLC blocks of the C = 64 / 128 levels on the fused spatial kernels + the tile-GEMM channel stage (A/B: 0 = per-operator launches)
    bool expand_wgrad_split = true;   // (A/B) the patch-expand weight gradient follows expand_split too
    bool act_bf16 = false;     // bf16 STORAGE mode (round 6): block-internal tensors of the JLC blocks and the full-resolution heads / their gradients are bf16 arrays (set_act_bf16; functional.set_precision("bf16"))
    int expand_split = 0;      // fp32 mode: patch-expand products as 3 (2 pieces) / 6 (3 pieces) bf16 MFMAs per pair instead of fp32 MFMAs (csrc/expand_mfma.hip, fp32-accurate)
    double in_eps = 1e-5, ln_eps = 1e-6;
} F;

inline void chk(int rc, const char* what) {
    if (rc != 0) {
        const char* m = vx_last_error();
        throw std::runtime_error(std::string(what) + " failed (rc=" + std::to_string(rc) + "): " + (m ? m : "?"));
    }
}
// ---- per-call profiling (bench.py's kernel pass): HIP events around every C-ABI call on the stream it is launched on --------------------------
struct ProfRec { const char* name; std::vector<long> key; hipEvent_t e0, e1; };
struct Prof { bool on = false; std::vector<ProfRec> recs; } PROF;
template <class T> inline void prof_key(std::vector<long>& k, const T& v) {
    if constexpr (std::is_integral_v<T> && !std::is_same_v<T, bool>) { if ((long)v > -(1L << 24) && (long)v < (1L << 24)) k.push_back((long)v); }
}
template <class T> inline void* last_arg(T&& t) { if constexpr (std::is_pointer_v<std::decay_t<T>>) return (void*)t; else return nullptr; }
template <class T, class... R> inline void* last_arg(T&&, R&&... r) { return last_arg(std::forward<R>(r)...); }
// VX(vx_entry, args...): call a C-ABI entry, turn its status into an exception; the last argument of every launching entry is the stream
template <class F, class... A> inline void vxcall(const char* name, F f, A&&... a) {
    if (!PROF.on) { chk(f(a...), name); return; }
    hipStream_t st = (hipStream_t)last_arg(a...);
    ProfRec r;
    r.name = name;
    (prof_key(r.key, a), ...);
    TORCH_CHECK(hipEventCreate(&r.e0) == hipSuccess && hipEventCreate(&r.e1) == hipSuccess, "hipEventCreate failed");
    TORCH_CHECK(hipEventRecord(r.e0, st) == hipSuccess, "hipEventRecord failed");
    chk(f(a...), name);
    TORCH_CHECK(hipEventRecord(r.e1, st) == hipSuccess, "hipEventRecord failed");
    PROF.recs.push_back(std::move(r));
}
#define VX(fn, ...) vxcall(#fn, fn, __VA_ARGS__)
// VXR: entries that answer 1 for "shape not covered, use the other kernel" (nothing launched): 0 / 1 are returned, anything else throws; recorded like VX
template <class F, class... A> inline int vxcall_rc(const char* name, F f, A&&... a) {
    if (!PROF.on) { const int rc = f(a...); if (rc != 0 && rc != 1) chk(rc, name); return rc; }
    hipStream_t st = (hipStream_t)last_arg(a...);
    ProfRec r;
    r.name = name;
    (prof_key(r.key, a), ...);
    TORCH_CHECK(hipEventCreate(&r.e0) == hipSuccess && hipEventCreate(&r.e1) == hipSuccess, "hipEventCreate failed");
    TORCH_CHECK(hipEventRecord(r.e0, st) == hipSuccess, "hipEventRecord failed");
    const int rc = f(a...);
    if (rc != 0 && rc != 1) chk(rc, name);
    TORCH_CHECK(hipEventRecord(r.e1, st) == hipSuccess, "hipEventRecord failed");
    if (rc == 0) PROF.recs.push_back(std::move(r));
    else { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    return rc;
}
#define VXR(fn, ...) vxcall_rc(#fn, fn, __VA_ARGS__)
inline const float* fp(const Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
inline float* mp(Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
inline Tensor contig(const Tensor& t) { return t.is_contiguous() ? t : t.contiguous(); }
inline void check_in(const Tensor& x, const char* what) {
    TORCH_CHECK(x.is_cuda(), "veloxseg_amd.", what, ": input is on ", x.device(), "; the VeloxSeg hot path runs only on an MI355X (HIP kernels, no CPU fallback)");
    TORCH_CHECK(x.scalar_type() == at::kFloat, "veloxseg_amd.", what, ": expected float32");
    TORCH_CHECK(x.device().index() == c10::hip::current_device(), "veloxseg_amd.", what, ": tensor lives on cuda:", (int)x.device().index(), " but the current device is cuda:",
                (int)c10::hip::current_device(), "; kernels launch on the current device's stream -- wrap the call in torch.cuda.device(tensor.device)");
}
inline float* grad_ptr(const Tensor& p) {          // running gradient buffer of a parameter (created zeroed on first use), as functional.grad_buf
    if (!p.defined() || !p.requires_grad()) return nullptr;
    if (!p.grad().defined()) const_cast<Tensor&>(p).mutable_grad() = at::zeros_like(p);
    return p.grad().data_ptr<float>();
}

// ------------------------------------------------------------------------------------------------------------------- weight-gradient side stream
// In the backward pass only the INPUT gradient of a convolution is on the critical path (the previous layer waits for it); its weight gradient
// is needed by the optimizer alone.  When enabled (TrainEngine turns it on around loss.backward() and joins before the all-reduce / AdamW),
// the weight-gradient launches are DEFERRED: each is queued as a closure that keeps its tensors alive; every VX_WG_FLUSH closures the queue is
// launched on one side stream behind a single event per launching stream (an event record + wait pair costs ~10 us of host time on this runtime,
// a pair per convolution made the step slower), so the next layers' backward no longer queues behind ~4 ms of weight-gradient kernels.  The
// closures (and with them dy, x and the temporaries) are released only after the final join, i.e. after the joining stream waits for the side
// stream, which is what makes the caching allocator's reuse of that memory safe without record_stream.
#define VX_WG_FLUSH 8
struct WgradSide {
    bool enabled = false;
    bool same = false;                 // deferral WITHOUT a side stream: the closures are launched at the join, each on the stream it was submitted from
    c10::optional<c10::hip::HIPStreamMasqueradingAsCUDA> stream;      // torch on ROCm presents HIP streams with the CUDA device type
    hipEvent_t ev[64] = {};
    unsigned next = 0;
    std::vector<std::pair<void*, std::function<void(void*)>>> pending;      // (stream the operands were produced on, launch closure)
    std::vector<int> pending_dev;                                            // device of each pending closure
    std::vector<std::function<void(void*)>> done;                            // launched, kept alive until the final join
} WG;

struct AttnFoldPool { std::vector<Tensor> delta, keep; std::vector<float*> dtab; std::vector<VxPwaPlan> plans; };
inline AttnFoldPool& attn_fold_pool(void* stream, int B, int M) {
    static std::map<std::tuple<void*, int, int>, AttnFoldPool> pools;
    return pools[std::make_tuple(stream, B, M)];
}
inline hipStream_t wg_stream(int dev) {
    if (!WG.stream.has_value()) {
        WG.stream = c10::hip::getStreamFromPoolMasqueradingAsCUDA(false, (c10::DeviceIndex)dev);
        for (auto& e : WG.ev) TORCH_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess, "hipEventCreate failed");
    }
    return WG.stream->stream();
}
inline void wg_order(hipStream_t after, hipStream_t waiter) {           // `waiter` waits for everything enqueued on `after` so far
    hipEvent_t e = WG.ev[WG.next++ & 63];
    TORCH_CHECK(hipEventRecord(e, after) == hipSuccess && hipStreamWaitEvent(waiter, e, 0) == hipSuccess, "wgrad stream ordering failed");
}
inline void wg_flush(int dev) {
    if (WG.pending.empty()) return;
    hipStream_t side = wg_stream(dev);
    void* seen[8]; int ns = 0;
    for (auto& p : WG.pending) {                                         // one event per distinct producing stream
        bool dup = false;
        for (int i = 0; i < ns; ++i) dup = dup || seen[i] == p.first;
        if (!dup) { wg_order((hipStream_t)p.first, side); if (ns < 8) seen[ns++] = p.first; }
    }
    {
        // the closures allocate their temporaries when they are launched: make the side stream current, so that the caching allocator hands out
        // blocks whose earlier users are ordered before the side stream's kernels (a block of another stream's pool may still be in use there)
        c10::hip::HIPStreamGuardMasqueradingAsCUDA guard(*WG.stream);
        for (auto& p : WG.pending) { p.second((void*)side); WG.done.push_back(std::move(p.second)); }
    }
    WG.pending.clear(); WG.pending_dev.clear();
}
// run `launch(stream)` now on `cur`, or queue it for the side stream
inline void wgrad_submit(void* cur, int dev, std::function<void(void*)> launch) {
    if (!WG.enabled) { launch(cur); return; }
    WG.pending.emplace_back(cur, std::move(launch));
    WG.pending_dev.push_back(dev);
    if (!WG.same && WG.pending.size() >= VX_WG_FLUSH) wg_flush(dev);
}

// ------------------------------------------------------------------------------------------------------------------- convolution
// bf16 storage mode: the caller of the NEXT patch-expand forward on this thread asks for a 16-bit output (m.conv_h); consumed by conv_fwd_impl
thread_local bool CONV_OUT_H16 = false;
struct ConvState {
    Tensor x, x2, w, b;
    int B = 0, C1 = 0, Cin = 0, D = 0, H = 0, W = 0, Cout = 0, K = 0, S = 0, P = 0, G = 0, ps = 0;
    bool pw = false, s1 = false, patch = false, cm = false;      // cm: strided dense conv on MFMA (csrc/conv_mfma.hip)
    Tensor xmax;                                                 // stem: the bits of max |x| left by the forward kernel for the weight gradient (one int32)
    Tensor wt_bwd_pre;                                           // patch-expand: the input gradient's weight image when it was built ahead (EXPAND_PRE)
    Tensor wt_fwd;                                               // patch-expand, fp16-piece mode: the forward's weight-image workspace (its tail holds the weight tensor's scale word for the backward)
    bool patch_fused = false;                                    // patch embedding read in place (vx_patch_embed_*): x is the network input, not a patchified copy
};

// (round 6) both weight images of a patch-expand layer built ahead of its forward (fp16-piece mode): keyed by the weight's data pointer, consumed once by conv_fwd_impl
struct ExpandPreImg { Tensor wt_fwd, wt_bwd; int Cc = 0; bool keep = false; int64_t epoch = 0, ver = 0; };      // keep / epoch / ver: as JlcPreImg below (inference)
static int64_t vx_weights_epoch_now();
static std::unordered_map<const void*, ExpandPreImg> EXPAND_PRE;
Tensor conv_fwd_impl(ConvState& st, const Tensor& x_in, const Tensor& x2_in, const Tensor& w, const Tensor& b, int K, int S, int P, int G, int ps,
                     bool x_requires_grad, void* stream) {
    check_in(x_in, "conv3d");
    // a channel slice of a wider contiguous tensor (the per-modality chunks of the network input): the patch-embedding path reads it in place
    const bool slice = x_in.dim() == 5 && !x_in.is_contiguous() && !x2_in.defined() && x_in.stride(4) == 1 && x_in.stride(3) == x_in.size(4) &&
                       x_in.stride(2) == x_in.size(3) * x_in.size(4) && (x_in.size(1) == 1 || x_in.stride(1) == x_in.size(2) * x_in.size(3) * x_in.size(4)) &&
                       F.use_patchify && K == S && (K == 2 || K == 4) && P == 0 && G == 1 && ps == 1 && !x_requires_grad && x_in.stride(0) % 4 == 0 &&
                       x_in.size(2) % K == 0 && x_in.size(3) % K == 0 && x_in.size(4) % K == 0 && (x_in.size(1) * K * K * K) % 4 == 0 && (x_in.storage_offset() % 4) == 0;
    Tensor x = slice ? x_in : contig(x_in), x2 = x2_in.defined() ? contig(x2_in) : Tensor();
    const int B = x.size(0), C1 = x.size(1), D = x.size(2), H = x.size(3), W = x.size(4);
    const int Cin = C1 + (x2.defined() ? (int)x2.size(1) : 0), Cout = w.size(0);
    TORCH_CHECK(w.size(1) * G == Cin && w.size(2) == K, "conv3d: weight shape does not match the input");
    const int Do = (D + 2 * P - K) / S + 1, Ho = (H + 2 * P - K) / S + 1, Wo = (W + 2 * P - K) / S + 1;
    // 16-bit output: only where the bf16-operand patch-expand kernel runs (its _h entry stores bf16); any other convolution ignores the request
    const bool want_h16 = CONV_OUT_H16;
    CONV_OUT_H16 = false;
    const bool y_h16 = want_h16 && F.act_bf16 && F.bf16_expand && F.use_expand_mfma && !x2_in.defined() && ps == 4 && K == 3 && S == 1 && P == 1 && Cin == 16 && G == 1 && Cout % 64 == 0 &&
                       D % 4 == 0 && H % 4 == 0 && W % 4 == 0 && F.use_s1;
    auto yopt = y_h16 ? x.options().dtype(at::kBFloat16) : x.options();
    Tensor y = ps == 1 ? at::empty({B, Cout, Do, Ho, Wo}, yopt) : at::empty({B, Cout / (ps * ps * ps), Do * ps, Ho * ps, Wo * ps}, yopt);
    const long V = (long)D * H * W;
    st.pw = (K == 1 && S == 1 && P == 0 && G == 1 && ps == 1 && Cin % 4 == 0 && C1 % 4 == 0);
    st.s1 = (!x2.defined() && S == 1 && (K == 3 || K == 5) && P == K / 2 && (Cout / G) % 4 == 0 && (Cin / G) % 4 == 0 && F.use_s1);
    st.patch = (F.use_patchify && K == S && (K == 2 || K == 4) && P == 0 && G == 1 && ps == 1 && !x2.defined() && !x_requires_grad &&
                D % K == 0 && H % K == 0 && W % K == 0 && (Cin * K * K * K) % 4 == 0);
    st.cm = (F.use_conv_mfma && !x2.defined() && !st.patch && S >= 2 && vx_conv_mfma_ok(Cin, Cout, D, H, W, K, S, P, G, ps) == 1);
    st.w = w; st.b = b;
    st.B = B; st.C1 = C1; st.Cin = Cin; st.D = D; st.H = H; st.W = W; st.Cout = Cout; st.K = K; st.S = S; st.P = P; st.G = G; st.ps = ps;
    if (st.patch) {
        const int Ck = Cin * K * K * K;
        const long Vo = (long)Do * Ho * Wo;
        // (round 6) no patchified copy where the fused kernels take the layer: the product and (later) the weight gradient gather the patches from x by address
        st.patch_fused = vx_patch_embed_ok(Cin, Cout, Do, Ho, Wo, K) == 1 && x.stride(0) % 4 == 0 && (reinterpret_cast<uintptr_t>(x.data_ptr()) & 15) == 0;
        if (st.patch_fused) {
            VX(vx_patch_embed_fwd, fp(x), (long)x.stride(0), fp(w), fp(b), mp(y), B, Cin, Cout, Do, Ho, Wo, stream);
            st.x = x;
            st.pw = false;
            return y;
        }
        // bf16 storage mode: the patchified copy of the network input (written once, read by this product and by its weight gradient) as a bf16 array on the large grids
        static const bool patch_h16 = !(getenv("VELOXSEG_BF16_PATCH") && getenv("VELOXSEG_BF16_PATCH")[0] == '0');      // (A/B)
        // (only where the product runs on the 16-byte-per-lane kernel vx_pw_fwd_v4_k -- at most 64 patch channels, one input channel per modality: with BraTS' 256
        // patch channels the one-voxel-per-thread kernel would issue 2-byte loads and the step gets SLOWER: brats128 808 vs 818 patches/s, same box, twice)
        const bool xs_h16 = patch_h16 && F.act_bf16 && Vo >= 16384 && Vo % 4 == 0 && (K == 2 || K == 4) && Cout % 16 == 0 && Ck <= 64 && Ck % 4 == 0 && Wo % 2 == 0;
        Tensor xs = at::empty({B, Ck, Do, Ho, Wo}, xs_h16 ? x.options().dtype(at::kBFloat16) : x.options());
        VX(vx_patchify_bs_h, fp(x), (long)x.stride(0), xs.data_ptr(), B, Cin, Do, Ho, Wo, K, (int)xs_h16, stream);
        if (Vo <= F.pw_mfma_max_v) VX(vx_pw_conv_mfma, fp(xs), nullptr, Ck, fp(w), 0, fp(b), mp(y), nullptr, 0, B, Cout, Ck, Ck, Vo, 0, stream);
        else VX(vx_pw_conv_fwd_h, (const void*)xs.data_ptr(), fp(w), fp(b), mp(y), B, Ck, Cout, Vo, (int)xs_h16, stream);
        st.x = xs;
        st.pw = false;
        return y;
    }
    if (st.pw && V <= F.pw_mfma_max_v) VX(vx_pw_conv_mfma, fp(x), fp(x2), C1, fp(w), 0, fp(b), mp(y), nullptr, 0, B, Cout, Cin, Cin, V, 0, stream);
    else if (st.pw) VX(vx_pw_conv_fwd, fp(x), fp(x2), C1, fp(w), fp(b), mp(y), B, Cin, Cout, V, stream);
    else if (st.s1) {
        int rc = 1;
        if (ps == 4 && K == 3 && Cin == 16 && G == 1 && Cout % 64 == 0 && F.use_expand_mfma) {       // patch-expand layer: MFMA tiles over an LDS halo
            const long wsn = std::max<long>((long)Cout * 16 * 27, F.expand_split ? (long)vx_expand_split_ws_floats(Cout / 64, F.expand_split) : 0);
            Tensor wt = at::empty({wsn}, x.options());
            if (F.bf16_expand) {
                rc = VXR(vx_expand_fwd_mfma_bf16_h, fp(x), fp(w), fp(b), mp(wt), y.data_ptr(), B, Cout / 64, D, H, W, (int)y_h16, stream);
                TORCH_CHECK(rc == 0 || !y_h16, "conv3d: the 16-bit patch-expand output needs the bf16-operand kernel");
            } else if (F.expand_split) {
                st.wt_bwd_pre = Tensor();
                bool pre = false;
                auto it = EXPAND_PRE.find(w.data_ptr());
                if (it != EXPAND_PRE.end()) {
                    const ExpandPreImg& q = it->second;
                    pre = F.expand_split == 22 && q.Cc == Cout / 64 && D % 4 == 0 && H % 4 == 0 && W % 4 == 0;
                    if (pre && q.keep) pre = q.epoch == vx_weights_epoch_now() && q.ver == (int64_t)w._version();
                    if (pre) { wt = q.wt_fwd; st.wt_bwd_pre = q.wt_bwd; }
                    if (!(pre && q.keep)) EXPAND_PRE.erase(it);
                }
                if (pre) rc = VXR(vx_expand_fwd_mfma_split_prepared, fp(x), fp(b), fp(wt), mp(y), B, Cout / 64, D, H, W, stream);
                else rc = VXR(vx_expand_fwd_mfma_split, fp(x), fp(w), fp(b), mp(wt), mp(y), B, Cout / 64, D, H, W, F.expand_split, stream);
                static const bool ew_on = !(getenv("VELOXSEG_EXPAND_EW_FWD") && getenv("VELOXSEG_EXPAND_EW_FWD")[0] == '0');      // (A/B)
                st.wt_fwd = (rc == 0 && F.expand_split == 22 && (ew_on || pre) && w.requires_grad()) ? wt : Tensor();
            }
            if (rc == 1) rc = vx_expand_fwd_mfma(fp(x), fp(w), fp(b), mp(wt), mp(y), B, Cout / 64, D, H, W, stream);
            if (rc != 0 && rc != 1) chk(rc, "vx_expand_fwd_mfma");
        }
        if (rc == 1) VX(vx_conv_s1, fp(x), fp(w), fp(b), mp(y), B, Cin, Cout, D, H, W, K, G, 0, 1, ps, 0, stream);
    }
    else if (st.cm) {
        Tensor ws = at::empty({(long)vx_conv_mfma_ws_floats(Cin, Cout, K, 0)}, x.options());
        static const bool mx_on = !(getenv("VELOXSEG_STEM_ABSMAX_FWD") && getenv("VELOXSEG_STEM_ABSMAX_FWD")[0] == '0');      // (A/B)
        st.xmax = Tensor();
        if (mx_on && w.requires_grad() && vx_conv_mfma_fwd_writes_absmax(Cin, Cout, D, H, W, K, S, P) == 1) st.xmax = at::empty({1}, x.options().dtype(at::kInt));
        VX(vx_conv_mfma_fwd_mx, fp(x), fp(w), fp(b), mp(y), mp(ws), st.xmax.defined() ? reinterpret_cast<unsigned*>(st.xmax.data_ptr()) : nullptr, B, Cin, D, H, W, Cout, K, S, P, stream);
    }
    else VX(vx_conv3d_fwd, fp(x), fp(x2), C1, fp(w), fp(b), mp(y), B, Cin, D, H, W, Cout, K, S, P, G, ps, stream);
    st.x = x; st.x2 = x2;
    return y;
}

// dx / dx2 are returned undefined when need_x is false; parameter gradients are accumulated into w.grad / b.grad
// acc_into (optional, only without a concat input): the input gradient is ADDED to this tensor in place and returned as dx
void conv_bwd_impl(ConvState& st, const Tensor& dy_in, bool need_x, Tensor& dx, Tensor& dx2, void* stream, const Tensor& acc_into = Tensor(), bool skip_bias = false) {
    Tensor dy = contig(dy_in);
    const bool dy_h16 = dy.scalar_type() == at::kBFloat16;           // bf16 storage mode: the gradient of a 16-bit patch-expand output
    if (dy_h16) {
        TORCH_CHECK(st.s1 && st.ps == 4 && st.K == 3 && st.Cin == 16 && st.G == 1 && F.use_expand_mfma && F.bf16_expand && st.D % 4 == 0 && st.H % 4 == 0 && st.W % 4 == 0 && !st.x2.defined(),
                    "conv3d backward: a bfloat16 gradient exists for the patch-expand layers in the bf16 mode only");
    }
    const int B = st.B, C1 = st.C1, Cin = st.Cin, D = st.D, H = st.H, W = st.W, Cout = st.Cout, K = st.K, S = st.S, P = st.P, G = st.G, ps = st.ps;
    const long V = (long)D * H * W;
    const Tensor& x = st.x; const Tensor& x2 = st.x2; const Tensor& w = st.w; const Tensor& b = st.b;
    const int dev = dy.device().index();
    if (st.patch) {
        if (w.requires_grad()) {
            float* dw = grad_ptr(w); float* db = grad_ptr(b);
            const int x_h16 = x.scalar_type() == at::kBFloat16 ? 1 : 0;          // (the patchified copy of the bf16 storage mode)
            if (st.patch_fused) {
                const long bs = x.stride(0);
                wgrad_submit(stream, dev, [=](void* s) {
                    VX(vx_patch_embed_bwd_weight, fp(x), bs, fp(dy), dw, db, B, Cin, Cout, D / K, H / K, W / K, s);
                });
                return;
            }
            wgrad_submit(stream, dev, [=](void* s) {
                VX(vx_pw_conv_bwd_weight_h, (const void*)x.data_ptr(), fp(dy), dw, db, B, Cin * K * K * K, Cout, (long)(D / K) * (H / K) * (W / K), x_h16, s);
            });
        }
        return;
    }
    if (need_x && st.pw && V <= F.pw_mfma_max_v && w.requires_grad() && (!WG.enabled || WG.same) && F.fuse_pw_bwd) {      // both gradients of a small 1x1 conv in one launch
        const int acc = (acc_into.defined() && !x2.defined()) ? 1 : 0;
        dx = acc ? acc_into : at::empty_like(x);
        if (x2.defined()) dx2 = at::empty_like(x2);
        VX(vx_pw_conv_bwd_fused, fp(dy), fp(w), fp(x), fp(x2), C1, mp(dx), mp(dx2), grad_ptr(w), skip_bias ? nullptr : grad_ptr(b), B, Cin, Cout, V, acc, stream);
        return;
    }
    if (need_x && st.pw && V > F.pw_mfma_max_v && w.requires_grad() && (!WG.enabled || WG.same) && F.fuse_pw_bwd) {     // the same at the large levels
        const int acc = (acc_into.defined() && !x2.defined()) ? 1 : 0;
        dx = acc ? acc_into : at::empty_like(x);
        if (x2.defined()) dx2 = at::empty_like(x2);
        VX(vx_pw_conv_bwd_fused_big, fp(dy), fp(w), fp(x), fp(x2), C1, mp(dx), mp(dx2), grad_ptr(w), skip_bias ? nullptr : grad_ptr(b), B, Cin, Cout, V, acc, stream);
        return;
    }
    if (need_x) {
        const int acc = (acc_into.defined() && !x2.defined()) ? 1 : 0;
        dx = acc ? acc_into : at::empty_like(x);
        if (x2.defined()) dx2 = at::empty_like(x2);
        if (st.pw && V <= F.pw_mfma_max_v) VX(vx_pw_conv_mfma, fp(dy), nullptr, 0, fp(w), 1, nullptr, mp(dx), mp(dx2), C1, B, Cin, Cout, Cin, V, acc, stream);
        else if (st.pw) VX(vx_pw_conv_bwd_data, fp(dy), fp(w), mp(dx), mp(dx2), C1, B, Cin, Cout, V, acc, stream);
        else if (st.s1 && ps == 4 && K == 3 && Cin == 16 && G == 1 && F.use_expand_mfma) {
            const long wsn = std::max<long>((long)Cout * 16 * 27, F.expand_split ? (long)vx_expand_split_ws_floats(Cout / 64, F.expand_split) : 0);
            Tensor wt = at::empty({wsn}, x.options());
            int rcb = 1;
            if (F.bf16_expand) {
                rcb = VXR(vx_expand_bwd_data_mfma_bf16_h, (const void*)dy.data_ptr(), fp(w), mp(wt), mp(dx), B, Cout / 64, D, H, W, acc, (int)dy_h16, stream);
                TORCH_CHECK(rcb == 0 || !dy_h16, "conv3d backward: the 16-bit gradient needs the bf16-operand kernel");
            } else if (F.expand_split) {
                const float* ew_fwd = (st.wt_fwd.defined() && F.expand_split == 22) ? st.wt_fwd.data_ptr<float>() + vx_expand_split_ew_offset(Cout / 64) : nullptr;
                if (ew_fwd && st.wt_bwd_pre.defined())
                    rcb = VXR(vx_expand_bwd_data_mfma_split_prepared, fp(dy), fp(st.wt_bwd_pre), ew_fwd, mp(dx), B, Cout / 64, D, H, W, acc, stream);
                else
                rcb = VXR(vx_expand_bwd_data_mfma_split_ew, fp(dy), fp(w), mp(wt), mp(dx), B, Cout / 64, D, H, W, acc, F.expand_split, ew_fwd, stream);
            }
            if (rcb == 1) VX(vx_expand_bwd_data_mfma, fp(dy), fp(w), mp(wt), mp(dx), B, Cout / 64, D, H, W, acc, stream);
        } else if (st.s1) VX(vx_conv_s1, fp(dy), fp(w), nullptr, mp(dx), B, Cout, Cin, D, H, W, K, G, 1, ps, 1, acc, stream);
        else if (st.cm) {
            Tensor ws = at::empty({(long)vx_conv_mfma_ws_floats(Cin, Cout, K, 1)}, x.options());
            VX(vx_conv_mfma_bwd_data, fp(dy), fp(w), mp(dx), mp(ws), B, Cin, D, H, W, Cout, K, S, P, acc, stream);
        }
        else VX(vx_conv3d_bwd_data, fp(dy), fp(w), nullptr, mp(dx), mp(dx2), C1, B, Cin, D, H, W, Cout, K, S, P, G, ps, acc, stream);
    }
    if (w.requires_grad()) {
        float* dw = grad_ptr(w);
        float* db = skip_bias ? nullptr : grad_ptr(b);       // skip_bias: the caller fused the bias gradient into the InstanceNorm backward
        const bool s1 = st.s1;
        const Tensor xmax = st.xmax;                         // (stem only: max |x| from the forward kernel)
        // the closure owns dy, x, x2 (by value) and its temporaries: with the side stream on, it is released only after the final join
        wgrad_submit(stream, dev, [=](void* s) {
            if (K == 1 && S == 1 && P == 0 && G == 1 && ps == 1) VX(vx_pw_conv_bwd_weight, fp(x), fp(x2), C1, fp(dy), dw, db, B, Cin, Cout, V, s);
            else if (F.use_gconv1 && K == 1 && S == 1 && P == 0 && G > 1 && ps == 1 && !x2.defined() && Cin == Cout && (Cin / G == 4 || Cin / G == 8 || Cin / G == 16) && V % 4 == 0)
                VX(vx_gconv1_bwd_weight, fp(x), fp(dy), dw, db, B, Cin, G, V, s);
            else if (s1 && ps == 4 && K == 3 && Cin == 16 && G == 1 && F.use_expand_mfma) {
                int rcw = 1;
                const int wns = F.bf16_expand ? 1 : (F.expand_split == 22 ? 3 : F.expand_split);       // bf16 opt-in mode: plain bf16 operands (one piece), fp32 accumulation
                if (wns >= 1 && F.expand_wgrad_split) {                   // fp32 mode: fp32-exact products on the bf16 pipe (expand_mfma.hip vx_expand_wgrad_split_k)
                    const long nws = vx_expand_wgrad_split_ws_floats(B, Cout / 64, D, H, W);
                    auto pws = std::make_shared<Tensor>(at::empty({nws}, x.options()));
                    WG.done.push_back([pws](void*) {});
                    rcw = VXR(vx_expand_wgrad_mfma_split_h, fp(x), (const void*)dy.data_ptr(), dw, db, mp(*pws), (long)nws, B, Cout / 64, D, H, W, wns, (int)dy_h16, s);
                }
                TORCH_CHECK(rcw == 0 || !dy_h16, "conv3d backward: the 16-bit gradient needs the split weight-gradient kernel");
                if (rcw == 1) {
                    auto xcl = std::make_shared<Tensor>(at::empty({(long)B * V * 16}, x.options()));
                    WG.done.push_back([xcl](void*) {});          // keeps the temporary alive as long as the launches of this pass
                    VX(vx_expand_wgrad_mfma, fp(x), mp(*xcl), fp(dy), dw, db, B, Cout / 64, D, H, W, s);
                }
            } else if (K == 7 && S == 4 && P == 3 && G == 1 && ps == 1 && !x2.defined() && F.use_down_mfma && vx_down_wgrad_ws_floats(B, Cin, D, H, W, Cout) > 0) {
                const int nws = vx_down_wgrad_ws_floats(B, Cin, D, H, W, Cout);          // stem DownConv: MFMA tiles + partial-sum slices
                auto ws = std::make_shared<Tensor>(at::empty({(long)nws}, x.options()));
                WG.done.push_back([ws](void*) {});
                VX(vx_down_wgrad_mfma_mx, fp(x), fp(dy), dw, db, mp(*ws), nws, xmax.defined() ? reinterpret_cast<const unsigned*>(xmax.data_ptr()) : nullptr, B, Cin, D, H, W, Cout, s);
            } else if (G == 1 && ps == 1 && S > 1 && !x2.defined() && vx_conv_wgrad_gather_ok(B, Cin, D, H, W, Cout, K, S, P) == 1) {
                VX(vx_conv_wgrad_gather_mfma, fp(x), fp(dy), dw, db, B, Cin, D, H, W, Cout, K, S, P, s);      // DownConv levels 2 - 4: gather-GEMM on the fp32 matrix pipe
            } else if (F.use_wgrad_ws) {
                const int nws = vx_conv3d_bwd_weight_ws_floats(B, Cin, D, H, W, Cout, K, S, P, G, ps);
                TORCH_CHECK(nws >= 0, "vx_conv3d_bwd_weight_ws_floats failed");
                auto ws = std::make_shared<Tensor>(nws > 0 ? at::empty({(long)nws}, x.options()) : Tensor());
                WG.done.push_back([ws](void*) {});
                VX(vx_conv3d_bwd_weight_tiled_ws, fp(x), fp(x2), C1, fp(dy), dw, db, mp(*ws), nws, B, Cin, D, H, W, Cout, K, S, P, G, ps, s);
            } else VX(vx_conv3d_bwd_weight_tiled, fp(x), fp(x2), C1, fp(dy), dw, db, B, Cin, D, H, W, Cout, K, S, P, G, ps, s);
        });
    }
    if (!WG.enabled) WG.done.clear();                        // immediate launches: nothing to keep
}

// ------------------------------------------------------------------------------------------------------------------- InstanceNorm
struct INState {
    std::vector<Tensor> ys, stats;
    int n = 0, act = 0;
    bool has_res = false;
    long BC = 0, V = 0;
};

Tensor in_fwd_impl(INState& st, const Tensor& res, bool act, const std::vector<Tensor>& ys_in, void* stream) {
    const int n = (int)ys_in.size();
    TORCH_CHECK(n >= 1 && n <= 3, "instance_norm: 1..3 inputs");
    check_in(ys_in[0], "instance_norm");
    st.ys.clear();
    for (auto& y : ys_in) st.ys.push_back(contig(y));
    const long BC = st.ys[0].size(0) * st.ys[0].size(1), V = st.ys[0].numel() / BC;
    st.n = n; st.act = act ? 1 : 0; st.has_res = res.defined(); st.BC = BC; st.V = V;
    Tensor out = at::empty_like(st.ys[0]);
    Tensor res_c = res.defined() ? contig(res) : Tensor();
    const float* yp[3] = {fp(st.ys[0]), n > 1 ? fp(st.ys[1]) : nullptr, n > 2 ? fp(st.ys[2]) : nullptr};
    st.stats.clear();
    if (F.use_in_row && V <= F.in_row_max) {
        Tensor sbuf = at::empty({n, BC * 2}, st.ys[0].options());
        for (int k = 0; k < n; ++k) st.stats.push_back(sbuf[k]);
        float* sp[3] = {mp(st.stats[0]), n > 1 ? mp(st.stats[1]) : nullptr, n > 2 ? mp(st.stats[2]) : nullptr};
        VX(vx_in_row_fwd, yp[0], yp[1], yp[2], sp[0], sp[1], sp[2], n, st.act, fp(res_c), mp(out), BC, V, (float)F.in_eps, stream);
    } else if (F.in_split2) {      // long rows: one partial-sum launch for all inputs + one apply launch that folds the partials itself
        Tensor sbuf = at::empty({n, BC * 2}, st.ys[0].options());
        for (int k = 0; k < n; ++k) st.stats.push_back(sbuf[k]);
        Tensor part = at::empty({(long)n * BC * 32}, st.ys[0].options().dtype(at::kDouble));
        VX(vx_in_fwd_split, yp[0], yp[1], yp[2], mp(st.stats[0]), n > 1 ? mp(st.stats[1]) : nullptr, n > 2 ? mp(st.stats[2]) : nullptr, part.data_ptr<double>(),
                            n, st.act, fp(res_c), mp(out), BC, V, (float)F.in_eps, stream);
    } else {
        for (int k = 0; k < n; ++k) {
            Tensor s = at::empty({BC * 2}, st.ys[0].options());
            Tensor part = at::empty({BC * 32}, st.ys[0].options().dtype(at::kDouble));
            VX(vx_in_stats, yp[k], mp(s), part.data_ptr<double>(), BC, V, (float)F.in_eps, stream);
            st.stats.push_back(s);
        }
        VX(vx_in_apply_fwd, yp[0], yp[1], yp[2], fp(st.stats[0]), n > 1 ? fp(st.stats[1]) : nullptr, n > 2 ? fp(st.stats[2]) : nullptr, n, st.act, fp(res_c),
                            mp(out), BC, V, stream);
    }
    return out;
}

// grads[k] defined for every k with need[k]; the residual gradient is dout itself
inline Tensor sum3(const Tensor& a, const Tensor& b, const Tensor& c, void* stream);
// add0 (optional, single-input norms only): grads[0] = add0 + gradient (the sum with a residual-branch gradient of the same tensor, in the kernel's store)
std::vector<Tensor> in_bwd_impl(INState& st, const Tensor& dout_in, const std::vector<bool>& need, void* stream, float* const* dbs = nullptr, int C = 1,
                                const Tensor& add0 = Tensor()) {
    Tensor dout = contig(dout_in);
    Tensor addc = add0.defined() ? contig(add0) : Tensor();
    TORCH_CHECK(!addc.defined() || (st.n == 1 && !dbs), "in_bwd_impl: add0 only for single-input norms");
    const int n = st.n;
    std::vector<Tensor> grads(n);
    bool any = false;
    for (int k = 0; k < n; ++k) if (need[k]) { grads[k] = at::empty_like(st.ys[k]); any = true; }
    if (!any) return grads;
    if (addc.defined() && F.use_in_row && st.V <= F.in_row_max) {
        VX(vx_in_row_bwd_add, fp(dout), fp(st.ys[0]), fp(st.stats[0]), st.act, fp(addc), mp(grads[0]), st.BC, st.V, stream);
    } else if (F.use_in_row && st.V <= F.in_row_max) {
        VX(vx_in_row_bwd_db, fp(dout), fp(st.ys[0]), n > 1 ? fp(st.ys[1]) : nullptr, n > 2 ? fp(st.ys[2]) : nullptr, fp(st.stats[0]), n > 1 ? fp(st.stats[1]) : nullptr,
                             n > 2 ? fp(st.stats[2]) : nullptr, n, st.act, mp(grads[0]), n > 1 ? mp(grads[1]) : nullptr, n > 2 ? mp(grads[2]) : nullptr,
                             dbs ? dbs[0] : nullptr, (dbs && n > 1) ? dbs[1] : nullptr, (dbs && n > 2) ? dbs[2] : nullptr, C, st.BC, st.V, stream);
    } else if (F.in_split2 && !dbs) {
        Tensor part = at::empty({(long)n * st.BC * 32}, dout.options().dtype(at::kDouble));
        VX(vx_in_bwd_split, fp(dout), fp(st.ys[0]), n > 1 ? fp(st.ys[1]) : nullptr, n > 2 ? fp(st.ys[2]) : nullptr, fp(st.stats[0]), n > 1 ? fp(st.stats[1]) : nullptr,
                            n > 2 ? fp(st.stats[2]) : nullptr, part.data_ptr<double>(), n, st.act, mp(grads[0]), n > 1 ? mp(grads[1]) : nullptr,
                            n > 2 ? mp(grads[2]) : nullptr, fp(addc), st.BC, st.V, stream);
    } else {
        for (int k = 0; k < n; ++k) {
            if (!need[k]) continue;
            Tensor ws = at::empty({st.BC * 2}, dout.options());
            Tensor part = at::empty({st.BC * 32}, dout.options().dtype(at::kDouble));
            VX(vx_in_bwd_db, fp(dout), fp(st.ys[k]), fp(st.stats[k]), st.act, mp(ws), part.data_ptr<double>(), mp(grads[k]), st.BC, st.V, dbs ? dbs[k] : nullptr, C, stream);
        }
        if (addc.defined()) grads[0] = sum3(addc, grads[0], Tensor(), stream);
    }
    return grads;
}

// ------------------------------------------------------------------------------------------------------------------- LayerNorm, GELU, residual
struct LNState { Tensor x, g, bt; };

Tensor ln_fwd_impl(LNState& st, const Tensor& x_in, const Tensor& g, const Tensor& bt, void* stream) {
    check_in(x_in, "layer_norm");
    st.x = contig(x_in); st.g = g; st.bt = bt;
    Tensor out = at::empty_like(st.x);
    const int B = st.x.size(0), C = st.x.size(1);
    VX(vx_ln_cf_fwd, fp(st.x), fp(g), fp(bt), mp(out), B, C, st.x.numel() / ((long)B * C), (float)F.ln_eps, stream);
    return out;
}
Tensor ln_bwd_impl(LNState& st, const Tensor& dout_in, void* stream, const Tensor& add = Tensor()) {
    Tensor dout = contig(dout_in);
    const int B = st.x.size(0), C = st.x.size(1);
    const long V = st.x.numel() / ((long)B * C);
    Tensor dx = at::empty_like(st.x);
    Tensor ws = at::empty({2 * (long)B * V}, st.x.options());
    if (WG.enabled && (st.g.requires_grad() || st.bt.requires_grad())) {      // deferral on: the parameter gradients leave the chain (a closure that owns x, dout, ws)
        Tensor addc = add.defined() ? contig(add) : Tensor();
        VX(vx_ln_cf_bwd_data, fp(st.x), fp(st.g), fp(dout), fp(addc), mp(dx), mp(ws), B, C, V, (float)F.ln_eps, stream);
        Tensor x = st.x;
        float* dg = grad_ptr(st.g); float* db = grad_ptr(st.bt);
        wgrad_submit(stream, dx.device().index(), [=](void* s) { VX(vx_ln_cf_bwd_param, fp(x), fp(dout), fp(ws), dg, db, B, C, V, s); });
        return dx;
    }
    if (add.defined()) {
        Tensor addc = contig(add);
        VX(vx_ln_cf_bwd_add, fp(st.x), fp(st.g), fp(dout), fp(addc), mp(dx), grad_ptr(st.g), grad_ptr(st.bt), mp(ws), B, C, V, (float)F.ln_eps, stream);
    } else VX(vx_ln_cf_bwd, fp(st.x), fp(st.g), fp(dout), mp(dx), grad_ptr(st.g), grad_ptr(st.bt), mp(ws), B, C, V, (float)F.ln_eps, stream);
    return dx;
}

struct GeluState { Tensor a; double p = 0; int64_t site = 0; const void* rs = nullptr; };
Tensor gelu_fwd_impl(GeluState& st, const Tensor& a_in, double p, int64_t site, const void* rs, void* stream) {
    check_in(a_in, "gelu");
    st.a = contig(a_in); st.p = p; st.site = site; st.rs = p > 0 ? rs : nullptr;
    Tensor h = at::empty_like(st.a);
    VX(vx_gelu_drop_fwd, fp(st.a), mp(h), st.a.numel(), st.rs, (unsigned long long)site, (float)p, stream);
    return h;
}
Tensor gelu_bwd_impl(GeluState& st, const Tensor& dh_in, void* stream) {
    Tensor dh = contig(dh_in);
    Tensor da = at::empty_like(st.a);
    VX(vx_gelu_drop_bwd, fp(dh), fp(st.a), mp(da), st.a.numel(), st.rs, (unsigned long long)st.site, (float)st.p, stream);
    return da;
}

// "1x1 conv -> GELU -> dropout" in one launch (GELU in the conv epilogue) and its mirror "input gradient of the next 1x1 conv -> GELU backward";
// fills the same ConvState / GeluState as the two separate operators, so the weight gradients go through conv_bwd_impl unchanged.
bool pw_gelu_fusable(const Tensor& x, const Tensor& w) {
    return x.dim() == 5 && w.size(2) == 1 && w.size(1) == x.size(1) && x.size(1) % 4 == 0 && (x.numel() / (x.size(0) * x.size(1))) % 4 == 0;
}
Tensor pw_gelu_fwd_impl(ConvState& c, GeluState& g, const Tensor& x_in, const Tensor& w, const Tensor& b, double p, int64_t site, const void* rs, void* stream) {
    check_in(x_in, "conv3d");
    Tensor x = contig(x_in);
    const int B = x.size(0), Cin = x.size(1), D = x.size(2), H = x.size(3), W = x.size(4), Cout = w.size(0);
    const long V = (long)D * H * W;
    c.x = x; c.x2 = Tensor(); c.w = w; c.b = b;
    c.B = B; c.C1 = Cin; c.Cin = Cin; c.D = D; c.H = H; c.W = W; c.Cout = Cout; c.K = 1; c.S = 1; c.P = 0; c.G = 1; c.ps = 1;
    c.pw = true; c.s1 = false; c.patch = false;
    Tensor a = at::empty({B, Cout, D, H, W}, x.options()), h = at::empty({B, Cout, D, H, W}, x.options());
    g.a = a; g.p = p; g.site = site; g.rs = p > 0 ? rs : nullptr;
    VX(vx_pw_conv_gelu_fwd, fp(x), fp(w), fp(b), mp(a), mp(h), B, Cin, Cout, V, V <= F.pw_mfma_max_v ? 1 : 0, g.rs, (unsigned long long)site, (float)p, stream);
    return h;
}
// c2: the conv that consumed h; returns da (gradient at the pre-activation) and accumulates c2's weight / bias gradients
Tensor pw_gelu_bwd_impl(ConvState& c2, GeluState& g, const Tensor& dz_in, void* stream) {
    Tensor dz = contig(dz_in);
    const long V = (long)c2.D * c2.H * c2.W;
    Tensor da = at::empty_like(g.a);
    VX(vx_pw_conv_gelu_bwd_data, fp(dz), fp(c2.w), fp(g.a), mp(da), c2.B, c2.Cin, c2.Cout, V, V <= F.pw_mfma_max_v ? 1 : 0, g.rs, (unsigned long long)g.site,
                                 (float)g.p, stream);
    Tensor t1, t2;
    conv_bwd_impl(c2, dz, false, t1, t2, stream);          // parameter gradients only
    return da;
}

struct AxpyState { double alpha = 1, p = 0; int64_t site = 0; const void* rs = nullptr; bool has_x = false; };
Tensor axpy_fwd_impl(AxpyState& st, const Tensor& x, const Tensor& z_in, double alpha, double p, int64_t site, const void* rs, void* stream) {
    check_in(z_in, "residual_dropout");
    Tensor z = contig(z_in), xc = x.defined() ? contig(x) : Tensor();
    st.alpha = alpha; st.p = p; st.site = site; st.rs = p > 0 ? rs : nullptr; st.has_x = x.defined();
    Tensor out = at::empty_like(z);
    VX(vx_axpy_drop_fwd, fp(xc), fp(z), mp(out), (float)alpha, z.numel(), st.rs, (unsigned long long)site, (float)p, stream);
    return out;
}
// out = alpha * res + drop(conv1x1(x)) in one launch (residual + dropout in the conv epilogue); fills the ConvState / AxpyState of the two operators
Tensor pw_res_fwd_impl(ConvState& c, AxpyState& r, const Tensor& x_in, const Tensor& w, const Tensor& b, const Tensor& res_in, double alpha, double p, int64_t site,
                       const void* rs, void* stream) {
    check_in(x_in, "conv3d");
    Tensor x = contig(x_in), res = contig(res_in);
    const int B = x.size(0), Cin = x.size(1), D = x.size(2), H = x.size(3), W = x.size(4), Cout = w.size(0);
    const long V = (long)D * H * W;
    c.x = x; c.x2 = Tensor(); c.w = w; c.b = b;
    c.B = B; c.C1 = Cin; c.Cin = Cin; c.D = D; c.H = H; c.W = W; c.Cout = Cout; c.K = 1; c.S = 1; c.P = 0; c.G = 1; c.ps = 1;
    c.pw = true; c.s1 = false; c.patch = false;
    r.alpha = alpha; r.p = p; r.site = site; r.rs = p > 0 ? rs : nullptr; r.has_x = true;
    Tensor out = at::empty({B, Cout, D, H, W}, x.options());
    TORCH_CHECK(res.numel() == out.numel(), "residual shape does not match the conv output");
    VX(vx_pw_conv_res_fwd, fp(x), fp(w), fp(b), fp(res), mp(out), B, Cin, Cout, V, V <= F.pw_mfma_max_v ? 1 : 0, (float)alpha, r.rs, (unsigned long long)site, (float)p,
                           stream);
    return out;
}

// -> (dx or undefined, dz)
void axpy_bwd_impl(AxpyState& st, const Tensor& dout_in, bool need_x_in, Tensor& dx, Tensor& dz, void* stream) {
    Tensor dout = contig(dout_in);
    const bool need_x = st.has_x && need_x_in;
    if (st.p == 0.0 && (st.alpha == 1.0 || !need_x)) { if (need_x) dx = dout; dz = dout; return; }
    if (need_x) dx = st.alpha == 1.0 ? dout : at::empty_like(dout);
    dz = st.p == 0.0 ? dout : at::empty_like(dout);
    VX(vx_axpy_drop_bwd, fp(dout), (need_x && st.alpha != 1.0) ? mp(dx) : nullptr, st.p > 0 ? mp(dz) : nullptr, (float)st.alpha, dout.numel(), st.rs,
                         (unsigned long long)st.site, (float)st.p, stream);
}

inline Tensor sum3(const Tensor& a, const Tensor& b, const Tensor& c, void* stream) {
    Tensor ac = contig(a), bc = contig(b), cc = c.defined() ? contig(c) : Tensor();
    Tensor out = at::empty_like(ac);
    VX(vx_add, fp(ac), fp(bc), fp(cc), mp(out), ac.numel(), stream);
    return out;
}

// ------------------------------------------------------------------------------------------------------------------- composites
// state of the fused block kernels (jlc.hip + mlp.hip): what the backward pass recomputes from
struct JLCFusedState {
    Tensor x, y, o, stats_y, stats_o;            // y: (3, B, C, D, H, W) = the three conv outputs
    bool img_cl = false;                         // ... of csrc/jlc_cl.hip (coarse levels) instead
    int tz_pieces = 0;                           // the pieces mode `img` was built for (vx_jlc_tz_pieces() at forward time): the backward and the deferred weight gradients use it
    Tensor img;                                  // operand images of the three weight tensors (vx_jlc_tz_prep) when the convolutions ran on jlc_mfma.hip
    Tensor w1, w3, w5, b1, b3, b5, l1w, l1b, l2w, l2b;
    int B = 0, C = 0, G = 0, D = 0, H = 0, W = 0, R = 0, nch = 0;
    bool tile = false;                           // channel stage on the tile-GEMM kernels (pwa_fused.hip vx_inmlp_*: C = 64 / 128) instead of mlp.hip
    bool h16 = false;                            // bf16 storage mode: y, o (and dn, d_o, g in the backward pass) are bf16 arrays
    double p = 0; int64_t site = 0; const void* rs = nullptr;
};
struct JLCState {
    std::vector<ConvState> convs;
    INState in1, in2;
    ConvState c1, c2;
    GeluState g;
    AxpyState r;
    bool fused = false;
    bool blk = false;                            // the whole block ran on the fused kernels
    JLCFusedState f;
};

struct FFNState {
    LNState ln;
    ConvState c1, c2;
    GeluState g;
    AxpyState r;
    bool fused = false;
    bool blk = false;                            // LN + both 1x1 convs + residual in one launch (mlp.hip)
    Tensor y, gamma, beta, w1, b1, w2, b2;
    double p = 0; int64_t site1 = 0, site2 = 0; const void* rs = nullptr;
};

inline void* sp(int64_t v) { return reinterpret_cast<void*>(v); }

// ------------------------------------------------------------------------------------------------------------------- C++ autograd nodes
// The frequent single operators as torch::autograd::Function: forward and backward run without the interpreter (a python autograd.Function
// costs ~10 us per apply and ~12 us per backward call, on ~120 operators per step).  State travels in an IValue capsule on the node; the
// HIP stream is torch's current stream (the engine restores the forward's stream before it calls backward).
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;
template <class S> struct Holder : torch::CustomClassHolder { S s; };
template <class S> c10::intrusive_ptr<Holder<S>> put_state(AutogradContext* ctx) {
    auto h = c10::make_intrusive<Holder<S>>();
    ctx->saved_data["st"] = c10::IValue::make_capsule(h);
    return h;
}
template <class S> S& get_state(AutogradContext* ctx) {
    return c10::static_intrusive_pointer_cast<Holder<S>>(ctx->saved_data["st"].toCapsule())->s;
}
inline void* cur_stream(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

struct ConvFn : public torch::autograd::Function<ConvFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, const OptT& x2, const Tensor& w, const OptT& b, int64_t K, int64_t S, int64_t P, int64_t G, int64_t ps) {
        auto h = put_state<ConvState>(ctx);
        return conv_fwd_impl(h->s, x, x2.value_or(Tensor()), w, b.value_or(Tensor()), (int)K, (int)S, (int)P, (int)G, (int)ps, x.requires_grad(), cur_stream(x));
    }
    static variable_list backward(AutogradContext* ctx, variable_list g) {
        ConvState& st = get_state<ConvState>(ctx);
        Tensor dx, dx2;
        // needs_input_grad indexes the tensor arguments that were present (x, [x2], w, [b])
        conv_bwd_impl(st, g[0], ctx->needs_input_grad(0) || (st.x2.defined() && ctx->needs_input_grad(1)), dx, dx2, cur_stream(g[0]));
        ctx->saved_data.clear();
        return {dx, dx2, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

struct InstNormFn : public torch::autograd::Function<InstNormFn> {
    static Tensor forward(AutogradContext* ctx, const OptT& res, bool act, const Tensor& y0, const OptT& y1, const OptT& y2) {
        auto h = put_state<INState>(ctx);
        std::vector<Tensor> ys{y0};
        if (y1.has_value() && y1->defined()) ys.push_back(*y1);
        if (y2.has_value() && y2->defined()) ys.push_back(*y2);
        ctx->saved_data["has_res"] = res.has_value() && res->defined();
        return in_fwd_impl(h->s, res.value_or(Tensor()), act, ys, cur_stream(y0));
    }
    static variable_list backward(AutogradContext* ctx, variable_list g) {
        INState& st = get_state<INState>(ctx);
        const bool has_res = ctx->saved_data["has_res"].toBool();
        const int off = has_res ? 1 : 0;                      // edges: [res], y0, [y1], [y2]
        std::vector<bool> need(st.n);
        for (int k = 0; k < st.n; ++k) need[k] = ctx->needs_input_grad(off + k);
        auto grads = in_bwd_impl(st, g[0], need, cur_stream(g[0]));
        grads.resize(3);
        Tensor dres = (has_res && ctx->needs_input_grad(0)) ? contig(g[0]) : Tensor();
        ctx->saved_data.clear();
        return {dres, Tensor(), grads[0], grads[1], grads[2]};
    }
};

struct LayerNormFn : public torch::autograd::Function<LayerNormFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& gamma, const Tensor& beta) {
        auto h = put_state<LNState>(ctx);
        return ln_fwd_impl(h->s, x, gamma, beta, cur_stream(x));
    }
    static variable_list backward(AutogradContext* ctx, variable_list g) {
        Tensor dx = ln_bwd_impl(get_state<LNState>(ctx), g[0], cur_stream(g[0]));
        ctx->saved_data.clear();
        return {dx, Tensor(), Tensor()};
    }
};

struct GeluFn : public torch::autograd::Function<GeluFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& a, double p, int64_t site, int64_t rs) {
        auto h = put_state<GeluState>(ctx);
        return gelu_fwd_impl(h->s, a, p, site, sp(rs), cur_stream(a));
    }
    static variable_list backward(AutogradContext* ctx, variable_list g) {
        Tensor da = gelu_bwd_impl(get_state<GeluState>(ctx), g[0], cur_stream(g[0]));
        ctx->saved_data.clear();
        return {da, Tensor(), Tensor(), Tensor()};
    }
};

// out = alpha * res + drop(conv1x1(x)) as ONE node (PWA mix conv + residual dropout, PWA.py:377): forward one launch, backward = mask the incoming
// gradient once, then the fused 1x1 backward
struct PwResState { ConvState c; AxpyState r; };
struct PwResFn : public torch::autograd::Function<PwResFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& w, const OptT& b, const Tensor& res, double alpha, double p, int64_t site, int64_t rs) {
        auto h = put_state<PwResState>(ctx);
        return pw_res_fwd_impl(h->s.c, h->s.r, x, w, b.value_or(Tensor()), res, alpha, p, site, sp(rs), cur_stream(x));
    }
    static variable_list backward(AutogradContext* ctx, variable_list g) {
        PwResState& st = get_state<PwResState>(ctx);
        void* s_ = cur_stream(g[0]);
        const bool has_b = st.c.b.defined();
        const int res_edge = has_b ? 3 : 2;                              // edges: x, w, [b], res
        Tensor dres, dz, dx, dx2;
        axpy_bwd_impl(st.r, g[0], ctx->needs_input_grad(res_edge), dres, dz, s_);
        conv_bwd_impl(st.c, dz, ctx->needs_input_grad(0), dx, dx2, s_);
        ctx->saved_data.clear();
        return {dx, Tensor(), Tensor(), dres, Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

// q, k, v = three 1x1 convs of the same input as ONE node (PWA.py:291-298): the three input gradients are accumulated into one tensor by the kernels'
// accumulate mode instead of two autograd add launches per modality and level
struct QKVState { ConvState c[3]; };
struct QKVFn : public torch::autograd::Function<QKVFn> {
    static variable_list forward(AutogradContext* ctx, const Tensor& x, const Tensor& wq, const OptT& bq, const Tensor& wk, const OptT& bk, const Tensor& wv, const OptT& bv) {
        auto h = put_state<QKVState>(ctx);
        void* s_ = cur_stream(x);
        const Tensor* ws[3] = {&wq, &wk, &wv};
        const OptT* bs[3] = {&bq, &bk, &bv};
        variable_list out;
        for (int j = 0; j < 3; ++j) out.push_back(conv_fwd_impl(h->s.c[j], x, Tensor(), *ws[j], bs[j]->value_or(Tensor()), 1, 1, 0, 1, 1, x.requires_grad(), s_));
        return out;
    }
    static variable_list backward(AutogradContext* ctx, variable_list g) {
        QKVState& st = get_state<QKVState>(ctx);
        const bool need_x = ctx->needs_input_grad(0);
        Tensor dx;
        for (int j = 0; j < 3; ++j) {
            if (!g[j].defined()) continue;
            Tensor t1, t2;
            conv_bwd_impl(st.c[j], g[j], need_x, t1, t2, cur_stream(g[j]), dx);        // first defined gradient creates dx, the others accumulate into it
            if (need_x && !dx.defined()) dx = t1;
        }
        ctx->saved_data.clear();
        return {dx, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

struct AxpyFn : public torch::autograd::Function<AxpyFn> {
    static Tensor forward(AutogradContext* ctx, const OptT& x, const Tensor& z, double alpha, double p, int64_t site, int64_t rs) {
        auto h = put_state<AxpyState>(ctx);
        return axpy_fwd_impl(h->s, x.value_or(Tensor()), z, alpha, p, site, sp(rs), cur_stream(z));
    }
    static variable_list backward(AutogradContext* ctx, variable_list g) {
        Tensor dx, dz;
        AxpyState& st = get_state<AxpyState>(ctx);
        axpy_bwd_impl(st, g[0], st.has_x && ctx->needs_input_grad(0), dx, dz, cur_stream(g[0]));
        ctx->saved_data.clear();
        return {dx, dz, Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

// Paired-Window Attention core (gather -> attention -> scatter for all modalities, PWA.py:106-200,308-327) as ONE C++ autograd node: no interpreter
// between its ~6 forward / ~9 backward launches.  qkv = (q0, k0, v0, q1, k1, v1, ...), (B, nb*heads*c, g0, g1, g2) each.
// the dispatcher ops receive the {seed, step} tensor from the caller; the states keep a REFERENCE to it (not only its address), so a temporary the caller drops
// before backward stays alive and the backward regenerates the masks of the forward.  The slot is filled by op_pwa / op_jlc / op_ffn right before apply() and taken
// by the node's forward on the same thread.
thread_local Tensor RNG_KEEP;
inline Tensor take_rng_keep() { Tensor t = RNG_KEEP; RNG_KEEP = Tensor(); return t; }

struct PwaState {
    Tensor rs_keep;
    VxPwaPlan plan;
    Tensor tq, tk, tv, O, lse, tbl, iq, ik, iv, table, mbits;      // mbits: the forward's dropout keep bits (1 per score element), read by the one-pass backward
    std::vector<std::vector<int64_t>> shapes;
    int cq = 0, cv = 0, M = 0, B = 0;
    double p = 0; int64_t site = 0; const void* rs = nullptr;
};
struct PwaCoreFn : public torch::autograd::Function<PwaCoreFn> {
    static variable_list forward(AutogradContext* ctx, const Tensor& table, int64_t plan_ptr, int64_t cq, int64_t cv, double p_attn, int64_t site, int64_t rs,
                                 at::TensorList qkv) {
        auto h = put_state<PwaState>(ctx);
        PwaState& st = h->s;
        st.rs_keep = take_rng_keep();
        st.plan = *reinterpret_cast<const VxPwaPlan*>(plan_ptr);
        const VxPwaPlan* pp = &st.plan;
        const int M = (int)qkv.size() / 3;
        TORCH_CHECK(M >= 1 && (int)qkv.size() == 3 * M && M <= 4, "pwa_core: expected 3*M tensors, M <= 4");
        check_in(qkv[0], "pwa_attention");
        void* s_ = cur_stream(qkv[0]);
        const int B = qkv[0].size(0), hd = pp->heads, Nt = pp->Ntot, ML = M * pp->l;
        st.cq = (int)cq; st.cv = (int)cv; st.M = M; st.B = B; st.p = p_attn; st.site = site; st.rs = p_attn > 0 ? sp(rs) : nullptr; st.table = table;
        auto opt = qkv[0].options();
        st.tq = at::empty({B, hd, Nt, ML, cq}, opt); st.tk = at::empty({B, hd, Nt, ML, cq}, opt); st.tv = at::empty({B, hd, Nt, ML, cv}, opt);
        auto iopt = opt.dtype(at::kInt);
        st.iq = at::empty({B, hd, Nt, ML, cq}, iopt); st.ik = at::empty({B, hd, Nt, ML, cq}, iopt); st.iv = at::empty({B, hd, Nt, ML, cv}, iopt);
        std::vector<Tensor> keep;
        const float* srcs[12];
        for (int i = 0; i < 3 * M; ++i) { keep.push_back(contig(qkv[i])); srcs[i] = fp(keep.back()); st.shapes.push_back(keep.back().sizes().vec()); }
        VX(vx_pwa_gather_all_fwd, srcs, mp(st.tq), mp(st.tk), mp(st.tv), st.iq.data_ptr<int>(), st.ik.data_ptr<int>(), st.iv.data_ptr<int>(), pp, (int)cq, (int)cv, M, B, s_);
        st.O = at::empty_like(st.tv);
        st.lse = at::empty({B, hd, Nt, ML}, opt);
        st.tbl = contig(table);
        void* mb = nullptr;
        if (p_attn > 0 && vx_pwa_attn_mbits_useful(pp, B, M, (int)cq, (int)cv) == 1) {
            const int nmb = vx_pwa_attn_mbits_words(pp, B, M);
            TORCH_CHECK(nmb >= 0, "vx_pwa_attn_mbits_words failed");
            st.mbits = at::empty({(long)nmb}, opt.dtype(at::kShort));
            mb = st.mbits.data_ptr();
        }
        VX(vx_pwa_attn_fwd_mb, fp(st.tq), fp(st.tk), fp(st.tv), fp(st.tbl), mp(st.O), mp(st.lse), pp, B, M, (int)cq, (int)cv, st.rs, (unsigned long long)site, (float)p_attn, mb, s_);
        variable_list outs;
        float* optr[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int m = 0; m < M; ++m) {
            outs.push_back(at::empty({B, (long)pp->nb * hd * cv, pp->grid[0], pp->grid[1], pp->grid[2]}, opt));
            optr[m] = outs.back().data_ptr<float>();
        }
        VX(vx_pwa_scatter_fwd_all, fp(st.O), optr, pp, (int)cv, M, B, s_);      // every modality in one launch
        return outs;
    }
    static variable_list backward(AutogradContext* ctx, variable_list g) {
        PwaState& st = get_state<PwaState>(ctx);
        const VxPwaPlan* pp = &st.plan;
        const int M = st.M, B = st.B;
        Tensor ref;
        for (auto& t : g) if (t.defined()) { ref = t; break; }
        variable_list out(7 + 3 * M);
        if (!ref.defined()) { ctx->saved_data.clear(); return out; }
        void* s_ = cur_stream(ref);
        Tensor dO, delta_pre;                            // delta_pre: the attention backward's workspace when its replicas were zeroed by the scatter adjoint
        {
            std::vector<Tensor> gm(M);
            const float* gptr[4] = {nullptr, nullptr, nullptr, nullptr};
            bool all = true;
            for (int m = 0; m < M; ++m) { all = all && g[m].defined(); if (g[m].defined()) { gm[m] = contig(g[m]); gptr[m] = gm[m].data_ptr<float>(); } }
            if (all) {                                   // every modality in one launch per scale
                // (round 6) no fill launch in front: the adjoint kernels write every element themselves (vx_pwa_scatter_bwd_all_wz; 1 = not for this plan), and the same
                // launch zeroes the bias-gradient replicas of the attention backward below (one more 6 us launch off the chain)
                dO = at::empty_like(st.O);
                const int nws0 = vx_pwa_attn_bwd_ws_floats(pp, B, M), roff = vx_pwa_attn_bwd_rep_offset(pp, B, M);
                TORCH_CHECK(nws0 >= 0 && roff >= 0, "vx_pwa_attn_bwd_ws_floats failed");
                delta_pre = at::empty({(long)nws0}, st.tq.options());
                if (VXR(vx_pwa_scatter_bwd_all_wz, gptr, mp(dO), pp, st.cv, M, B, delta_pre.data_ptr<float>() + roff, (long)(nws0 - roff), s_) == 1) {
                    delta_pre = Tensor();
                    dO.zero_();
                    VX(vx_pwa_scatter_bwd_all, gptr, mp(dO), pp, st.cv, M, B, s_);
                }
            } else {
                dO = at::zeros_like(st.O);
                for (int m = 0; m < M; ++m)
                    if (gptr[m]) VX(vx_pwa_scatter_bwd, gptr[m], mp(dO), pp, st.cv, m, M, B, s_);
            }
        }
        Tensor dq = at::empty_like(st.tq), dk = at::empty_like(st.tk), dv = at::empty_like(st.tv);
        const int nws = vx_pwa_attn_bwd_ws_floats(pp, B, M);
        TORCH_CHECK(nws >= 0, "vx_pwa_attn_bwd_ws_floats failed");
        Tensor delta = delta_pre.defined() ? delta_pre : at::empty({(long)nws}, st.tq.options());
        if (delta_pre.defined()) vx_pwa_attn_bwd_mark_rep_zeroed(delta.data_ptr<float>());
        Tensor dtab_tmp;
        float* dtab = grad_ptr(st.table);
        if (!dtab) { dtab_tmp = at::zeros_like(st.tbl); dtab = dtab_tmp.data_ptr<float>(); }
        int rc_nf = 1;
        if (WG.enabled) {                  // deferral on: the fold of the bias-table gradient (a parameter gradient) leaves the chain
            rc_nf = vx_pwa_attn_bwd_nofold_mb(fp(st.tq), fp(st.tk), fp(st.tv), fp(st.tbl), fp(st.O), fp(st.lse), fp(dO), mp(dq), mp(dk), mp(dv), mp(delta), pp, B, M, st.cq, st.cv,
                                              st.rs, (unsigned long long)st.site, (float)st.p, st.mbits.defined() ? st.mbits.data_ptr() : nullptr, s_);
            if (rc_nf != 0 && rc_nf != 1) chk(rc_nf, "vx_pwa_attn_bwd_nofold");
            if (rc_nf == 0) {
                // the folds of the pass's attention backward calls (one per transformer level) are pooled per (stream, B, M) and leave in ONE launch
                // when the queue is joined: four launches of 2-27 blocks at the launch floor were 41 us at the very end of the encoder backward
                AttnFoldPool& pool = attn_fold_pool(s_, B, M);
                const bool first = pool.delta.empty();
                pool.delta.push_back(delta); pool.keep.push_back(dtab_tmp); pool.dtab.push_back(dtab); pool.plans.push_back(st.plan);
                if (first) {
                    const int dev = dq.device().index();
                    const int Bc = B, Mc = M;
                    wgrad_submit(s_, dev, [s_, Bc, Mc](void* s) {
                        AttnFoldPool p = std::move(attn_fold_pool(s_, Bc, Mc));
                        attn_fold_pool(s_, Bc, Mc) = AttnFoldPool();
                        for (size_t lo = 0; lo < p.delta.size(); lo += 8) {
                            const int cnt = (int)std::min<size_t>(8, p.delta.size() - lo);
                            const float* dl[8]; float* dt[8]; const VxPwaPlan* pl[8];
                            for (int j = 0; j < cnt; ++j) { dl[j] = p.delta[lo + j].data_ptr<float>(); dt[j] = p.dtab[lo + j]; pl[j] = &p.plans[lo + j]; }
                            VX(vx_pwa_attn_bwd_fold_many, dl, dt, pl, cnt, Bc, Mc, s);
                        }
                        WG.done.push_back([p](void*) {});        // delta / temporary tables stay alive until the final join
                    });
                }
            }
        }
        if (rc_nf == 1)
        VX(vx_pwa_attn_bwd_mb, fp(st.tq), fp(st.tk), fp(st.tv), fp(st.tbl), fp(st.O), fp(st.lse), fp(dO), mp(dq), mp(dk), mp(dv), dtab, mp(delta), pp, B, M, st.cq, st.cv,
           st.rs, (unsigned long long)st.site, (float)st.p, (const void*)(st.mbits.defined() ? st.mbits.data_ptr() : nullptr), s_);
        float* dsts[12];
        for (int i = 0; i < 3 * M; ++i) { Tensor t = at::empty(st.shapes[i], st.tq.options()); out[7 + i] = t; dsts[i] = t.data_ptr<float>(); }
        VX(vx_pwa_gather_all_bwd, fp(dq), fp(dk), fp(dv), st.iq.data_ptr<int>(), st.ik.data_ptr<int>(), st.iv.data_ptr<int>(), dsts, pp, st.cq, st.cv, M, B, s_);
        ctx->saved_data.clear();
        return out;
    }
};

// ------------------------------------------------------------------------------------------------------------------- small single-kernel nodes
// ConvTranspose3d(k=2, s=2) (conv_blocks.py:29-35), PatchMerging's space-to-depth (attention_utils.py:144-159), F.interpolate(trilinear, align_corners)
// (VeloxSeg.py:183) and the SDKT Gram matrix (loss.py:39-60) as C++ nodes: the same C-ABI calls as the python nodes of functional.py.
struct UpconvFn : public torch::autograd::Function<UpconvFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x_in, const Tensor& w, const OptT& b_, bool feeds_in) {
        check_in(x_in, "conv_transpose3d");
        Tensor x = contig(x_in), b = b_.value_or(Tensor());
        const int B = x.size(0), Ci = x.size(1), d = x.size(2), h = x.size(3), wd = x.size(4), Co = w.size(1);
        TORCH_CHECK(w.size(0) == Ci && w.size(2) == 2 && w.size(3) == 2 && w.size(4) == 2, "conv_transpose_k2s2: weight must be (Ci, Co, 2, 2, 2)");
        Tensor y = at::empty({B, Co, 2 * d, 2 * h, 2 * wd}, x.options());
        VX(vx_upconv_k2s2_fwd, fp(x), fp(w), fp(b), mp(y), B, Ci, Co, d, h, wd, cur_stream(x));
        ctx->saved_data["x"] = x; ctx->saved_data["w"] = w; ctx->saved_data["b"] = b; ctx->saved_data["feeds_in"] = feeds_in;
        return y;
    }
    static variable_list backward(AutogradContext* ctx, variable_list g) {
        Tensor x = ctx->saved_data["x"].toTensor(), w = ctx->saved_data["w"].toTensor();
        Tensor b = ctx->saved_data["b"].isTensor() ? ctx->saved_data["b"].toTensor() : Tensor();
        // the output feeds an InstanceNorm directly (UpConv, conv_blocks.py:29-35): its backward removes the per-(b, c) mean, so the bias gradient is zero
        // up to round-off -- as for the JLC spatial convs it is not computed (the zero gradient tensor still exists)
        const bool skip_db = ctx->saved_data["feeds_in"].toBool() && F.skip_in_bias;
        Tensor dy = contig(g[0]), dx;
        const int B = x.size(0), Ci = x.size(1), d = x.size(2), h = x.size(3), wd = x.size(4), Co = w.size(1);
        void* s_ = cur_stream(dy);
        if (ctx->needs_input_grad(0)) { dx = at::empty_like(x); VX(vx_upconv_k2s2_bwd_data, fp(dy), fp(w), mp(dx), B, Ci, Co, d, h, wd, s_); }
        float* dw = grad_ptr(w);
        float* db = grad_ptr(b);
        if (skip_db) db = nullptr;
        if (dw || db)
            wgrad_submit(s_, x.device().index(), [=](void* s) {     // the transposed conv's weight gradient = the stride-2 conv's with x and dy swapped
                if (dw && F.upconv_wgrad_mfma && vx_upconv_k2s2_wgrad_ok(Ci, Co)) VX(vx_upconv_k2s2_wgrad, fp(x), fp(dy), dw, B, Ci, Co, d, h, wd, s);
                else if (dw) VX(vx_conv3d_bwd_weight_tiled, fp(dy), nullptr, 0, fp(x), dw, nullptr, B, Co, 2 * d, 2 * h, 2 * wd, Ci, 2, 2, 0, 1, 1, s);
                if (db) VX(vx_channel_sum, fp(dy), db, B, Co, 8L * d * h * wd, s);
            });
        if (!WG.enabled) WG.done.clear();
        ctx->saved_data.clear();
        return {dx, Tensor(), Tensor(), Tensor()};
    }
};
struct S2DFn : public torch::autograd::Function<S2DFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x_in) {
        check_in(x_in, "space_to_depth");
        Tensor x = contig(x_in);
        const int B = x.size(0), C = x.size(1), D = x.size(2), H = x.size(3), W = x.size(4);
        TORCH_CHECK(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "space_to_depth2: even extents only");
        Tensor out = at::empty({B, 8 * C, D / 2, H / 2, W / 2}, x.options());
        VX(vx_space_to_depth2, fp(x), mp(out), B, C, D / 2, H / 2, W / 2, 0, cur_stream(x));
        ctx->saved_data["shape"] = x.sizes().vec();
        return out;
    }
    static variable_list backward(AutogradContext* ctx, variable_list g) {
        auto shp = ctx->saved_data["shape"].toIntVector();
        Tensor dy = contig(g[0]), dx = at::empty(shp, dy.options());
        VX(vx_space_to_depth2, fp(dy), mp(dx), (int)shp[0], (int)shp[1], (int)shp[2] / 2, (int)shp[3] / 2, (int)shp[4] / 2, 1, cur_stream(dy));
        return {dx};
    }
};
struct UpsampleFn : public torch::autograd::Function<UpsampleFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x_in, int64_t D, int64_t H, int64_t W) {
        check_in(x_in, "upsample_trilinear");
        Tensor x = contig(x_in);
        Tensor out = at::empty({x.size(0), x.size(1), D, H, W}, x.options());
        VX(vx_upsample_trilinear_fwd, fp(x), mp(out), (long)x.size(0) * x.size(1), (int)x.size(2), (int)x.size(3), (int)x.size(4), (int)D, (int)H, (int)W, cur_stream(x));
        ctx->saved_data["shape"] = x.sizes().vec();
        return out;
    }
    static variable_list backward(AutogradContext* ctx, variable_list g) {
        auto shp = ctx->saved_data["shape"].toIntVector();
        Tensor dy = contig(g[0]), dx = at::empty(shp, dy.options());
        const long BC = shp[0] * shp[1];
        const int d = shp[2], h = shp[3], w = shp[4], H = dy.size(3), W = dy.size(4);
        Tensor ws = at::empty({BC * d * ((long)H * W + (long)h * W)}, dy.options());
        VX(vx_upsample_trilinear_bwd, fp(dy), mp(dx), mp(ws), BC, d, h, w, (int)dy.size(2), H, W, cur_stream(dy));
        return {dx, Tensor(), Tensor(), Tensor()};
    }
};
struct GramFn : public torch::autograd::Function<GramFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x_in) {
        check_in(x_in, "gram");
        Tensor x = contig(x_in);
        const int B = x.size(0), C = x.size(1);
        Tensor G = at::empty({B, C, C}, x.options());
        VX(vx_gram_fwd, fp(x), mp(G), B, C, (long)(x.numel() / ((long)B * C)), cur_stream(x));
        ctx->saved_data["x"] = x;
        return G;
    }
    static variable_list backward(AutogradContext* ctx, variable_list g) {
        Tensor x = ctx->saved_data["x"].toTensor(), dG = contig(g[0]), dx = at::empty_like(x);
        const int B = x.size(0), C = x.size(1);
        VX(vx_gram_bwd, fp(x), fp(dG), mp(dx), B, C, (long)(x.numel() / ((long)B * C)), cur_stream(dG));
        ctx->saved_data.clear();
        return {dx};
    }
};

// (round 6) weight images prepared AHEAD of their block (engine.TrainEngine prefetches them on a side lane at the head of the encoder forward: the two or three 6 us
// preparation launches in front of every JLC block's convolutions leave the forward chains).  Keyed by the data pointer of the 1x1x1 weight; consumed once by jlc_fwd_f.
struct JlcPreImg { Tensor img; int kind = 0, pieces = 0, C = 0, G = 0; bool keep = false; int64_t epoch = 0, ver = 0; };      // kind 1: Toeplitz images (jlc_mfma.hip), 2: channels-last images (jlc_cl.hip)
// keep (inference: engine.TapedPredictor): the entry survives its use -- the weights do not change between forwards -- and is valid while the process-wide weights epoch
// (bumped by every optimisation step of a TrainEngine, whose fused AdamW writes through raw pointers) and the version counters of the three weight tensors (bumped by any
// in-place torch update: optimizers, load_state_dict) are those of the moment the images were built; an invalid entry is dropped and the images are built in place
static int64_t WEIGHTS_EPOCH = 0;
static int64_t vx_weights_epoch_now() { return WEIGHTS_EPOCH; }
static int64_t jlc_wver(const Tensor& a, const Tensor& b, const Tensor& c) { return (int64_t)a._version() + (int64_t)b._version() + (int64_t)c._version(); }
static std::unordered_map<const void*, JlcPreImg> JLC_PRE;
static int jlc_img_kind(int C, int G, int D, int H, int W) {
    if (!F.fuse_blocks || !F.jlc_tz || (C / G) % 4 != 0) return 0;
    if (vx_jlc_tz_ok(C, G, D, H, W)) return 1;
    if (vx_jlc_cl_ok(C, G, D, H, W)) return 2;
    return 0;
}
static std::pair<Tensor, std::shared_ptr<JLCState>> jlc_fwd_f(const Tensor& x, const std::vector<Tensor>& ws, const std::vector<Tensor>& bs, int G, const Tensor& l1w, const Tensor& l1b, const Tensor& l2w,
                        const Tensor& l2b, double p, int64_t site, int64_t rs, int64_t stream) {
        auto st = std::make_shared<JLCState>();
        void* s_ = sp(stream);
        const int n = (int)ws.size();
        if (F.fuse_blocks && n == 3 && x.dim() == 5 && ws[0].size(2) == 1 && ws[1].size(2) == 3 && ws[2].size(2) == 5) {
            const int B = x.size(0), C = x.size(1), D = x.size(2), H = x.size(3), W = x.size(4), R = l1w.size(0);
            const long V = (long)D * H * W;
            const bool mlp_ok = vx_mlp_supported(C, R, V) && C < F.tile_min_c, tile_ok = !mlp_ok && F.jlc_tile && V % 4 == 0 && vx_inmlp_ok(C, R, V);
            if ((C / G) % 4 == 0 && (mlp_ok || tile_ok) && bs[0].defined() && bs[1].defined() && bs[2].defined()) {
                check_in(x, "jlc");
                JLCFusedState& f = st->f;
                st->blk = true;
                f.x = contig(x);
                f.w1 = ws[0]; f.w3 = ws[1]; f.w5 = ws[2]; f.b1 = bs[0]; f.b3 = bs[1]; f.b5 = bs[2]; f.l1w = l1w; f.l1b = l1b; f.l2w = l2w; f.l2b = l2b;
                f.B = B; f.C = C; f.G = G; f.D = D; f.H = H; f.W = W; f.R = R; f.p = p; f.site = site; f.rs = p > 0 ? sp(rs) : nullptr;
                const long BC = (long)B * C;
                const bool tz = F.jlc_tz && vx_jlc_tz_ok(C, G, D, H, W);
                const bool cl = !tz && F.jlc_tz && vx_jlc_cl_ok(C, G, D, H, W);          // the coarse levels: channels-last implicit GEMM on the f16 pipe (csrc/jlc_cl.hip)
                const int nty = tz ? vx_jlc_tz_ntiles(C, G, D, H, W) : cl ? vx_jlc_cl_ntiles(C, G, D, H, W) : vx_jlc_ntiles(B, C, G, D, H, W);
                TORCH_CHECK(nty > 0, "vx_jlc_ntiles failed");
                f.nch = vx_jlc_nchunks(BC, V);
                auto dopt = f.x.options().dtype(at::kDouble);
                Tensor part_y = at::empty({3, BC, nty, 2}, dopt), part_o = at::empty({BC, (long)f.nch, 2}, dopt);
                // bf16 storage mode: the block-internal tensors as bf16 arrays where the whole chain has 16-bit instances (Toeplitz convs with plain bf16 operands + mlp.hip channel stage)
                f.tile = tile_ok;
                f.tz_pieces = tz ? vx_jlc_tz_pieces() : 0;
                f.h16 = F.act_bf16 && tz && !tile_ok && f.tz_pieces == 1 && F.jlc_wg_tz && vx_jlc_wgrad_tz_ok(C, G, D, H, W) == 1 && V > 64;
                auto iopt = f.h16 ? f.x.options().dtype(at::kBFloat16) : f.x.options();
                const long esz = f.h16 ? 2 : 4;
                f.y = at::empty({3, B, C, D, H, W}, iopt);
                char* yp = (char*)f.y.data_ptr();
                const long n1 = BC * V;
                void *y1 = yp, *y3 = yp + esz * n1, *y5 = yp + 2 * esz * n1;
                bool pre = false;
                {
                    auto it = JLC_PRE.find(f.w1.data_ptr());
                    if (it != JLC_PRE.end()) {
                        const JlcPreImg& q = it->second;
                        pre = q.C == C && q.G == G && ((tz && q.kind == 1 && q.pieces == f.tz_pieces) || (cl && q.kind == 2));
                        if (pre && q.keep) pre = q.epoch == WEIGHTS_EPOCH && q.ver == jlc_wver(f.w1, f.w3, f.w5);
                        if (pre) f.img = q.img;
                        if (!(pre && q.keep)) JLC_PRE.erase(it);           // (a stale or mismatching entry is dropped: the images are then built here as before)
                    }
                }
                if (tz) {
                    if (!pre) {
                        f.img = at::empty({(long)vx_jlc_tz_img_floats_ns(C, G, f.tz_pieces)}, f.x.options());
                        VX(vx_jlc_tz_prep_ns, fp(f.w1), fp(f.w3), fp(f.w5), mp(f.img), C, G, f.tz_pieces, s_);
                    }
                    VX(vx_jlc_tz_fwd_h, fp(f.x), fp(f.img), fp(f.b1), fp(f.b3), fp(f.b5), y1, y3, y5, part_y.data_ptr<double>(), B, C, G, D, H, W, f.tz_pieces, (int)f.h16, s_);
                } else if (cl) {
                    f.img_cl = true;
                    if (!pre) {
                        f.img = at::empty({(long)vx_jlc_cl_img_floats(C, G)}, f.x.options());
                        VX(vx_jlc_cl_prep, fp(f.w1), fp(f.w3), fp(f.w5), mp(f.img), C, G, s_);
                    }
                    VX(vx_jlc_cl_fwd, fp(f.x), fp(f.img), fp(f.b1), fp(f.b3), fp(f.b5), (float*)y1, (float*)y3, (float*)y5, part_y.data_ptr<double>(), B, C, G, D, H, W, s_);
                } else
                VX(vx_jlc_conv_fwd, fp(f.x), fp(f.w1), fp(f.w3), fp(f.w5), fp(f.b1), fp(f.b3), fp(f.b5), (float*)y1, (float*)y3, (float*)y5, part_y.data_ptr<double>(), B, C, G, D, H, W, s_);
                f.stats_y = at::empty({3, BC, 2}, f.x.options());
                f.stats_o = at::empty({BC, 2}, f.x.options());
                f.o = at::empty(f.x.sizes(), iopt);
                VX(vx_jlc_mid_fwd_h, fp(f.x), (const void*)y1, (const void*)y3, (const void*)y5, part_y.data_ptr<double>(), nty, mp(f.stats_y), f.o.data_ptr(), part_o.data_ptr<double>(), BC, V, (float)F.in_eps, (int)f.h16, s_);
                Tensor out = at::empty_like(f.x);
                if (tile_ok) VX(vx_inmlp_fwd, fp(f.o), part_o.data_ptr<double>(), f.nch, mp(f.stats_o), fp(l1w), fp(l1b), fp(l2w), fp(l2b), mp(out), B, C, R, V, (float)F.in_eps,
                                f.rs, (unsigned long long)site, (float)p, s_);
                else
                VX(vx_mlp_fwd_h, (const void*)f.o.data_ptr(), 0, part_o.data_ptr<double>(), f.nch, mp(f.stats_o), nullptr, nullptr, fp(l1w), fp(l1b), fp(l2w), fp(l2b), mp(out), B, C, R, V,
                               (float)F.in_eps, f.rs, 0ull, 0.0f, (unsigned long long)site, (float)p, (int)f.h16, s_);
                return {out, st};
            }
        }
        st->convs.resize(n);
        std::vector<Tensor> ys;
        for (int k = 0; k < n; ++k) {
            const int K = ws[k].size(2);
            ys.push_back(conv_fwd_impl(st->convs[k], x, Tensor(), ws[k], bs[k], K, 1, K / 2, G, 1, x.requires_grad(), s_));
        }
        Tensor o = in_fwd_impl(st->in1, x, true, ys, s_);
        Tensor nrm = in_fwd_impl(st->in2, Tensor(), false, {o}, s_);
        st->fused = F.fuse_gelu && pw_gelu_fusable(nrm, l1w) && l2w.size(1) % 4 == 0;
        Tensor h;
        if (st->fused) h = pw_gelu_fwd_impl(st->c1, st->g, nrm, l1w, l1b, 0.0, 0, nullptr, s_);
        else {
            Tensor a = conv_fwd_impl(st->c1, nrm, Tensor(), l1w, l1b, 1, 1, 0, 1, 1, true, s_);
            h = gelu_fwd_impl(st->g, a, 0.0, 0, nullptr, s_);
        }
        Tensor out;
        if (F.fuse_res && pw_gelu_fusable(h, l2w)) out = pw_res_fwd_impl(st->c2, st->r, h, l2w, l2b, o, 1.0, p, site, sp(rs), s_);
        else {
            Tensor z = conv_fwd_impl(st->c2, h, Tensor(), l2w, l2b, 1, 1, 0, 1, 1, true, s_);
            out = axpy_fwd_impl(st->r, o, z, 1.0, p, site, sp(rs), s_);
        }
        return {out, st};
    }

static Tensor jlc_bwd_f(std::shared_ptr<JLCState> st, const Tensor& dout_in, bool need_x, int64_t stream) {
        void* s_ = sp(stream);
        if (st->blk) {
            JLCFusedState& f = st->f;
            Tensor dout = contig(dout_in);
            const int B = f.B, C = f.C, G = f.G, D = f.D, H = f.H, W = f.W, R = f.R;
            const long V = (long)D * H * W, BC = (long)B * C, n1 = BC * V;
            const int npd = f.tile ? vx_inmlp_tiles(V) : vx_mlp_bwd_nparts(B, C, V);
            auto iopt = f.h16 ? f.x.options().dtype(at::kBFloat16) : f.x.options();
            const long esz = f.h16 ? 2 : 4;
            const int h16 = f.h16 ? 1 : 0;
            Tensor dn = at::empty(f.x.sizes(), iopt), part_dn = at::empty({BC, (long)npd, 2}, f.x.options());
            if (f.tile) {
                // scratch operands of the two weight gradients (dW2 = dz h^T, dW1 = da nhat^T): one grouped launch, a sink of the pass
                Tensor sc_n = at::empty_like(f.x), sc_dz = at::empty_like(f.x), sc_h = at::empty({B, R, D, H, W}, f.x.options()), sc_da = at::empty({B, R, D, H, W}, f.x.options());
                VX(vx_inmlp_bwd, fp(f.o), fp(f.stats_o), fp(f.l1w), fp(f.l1b), fp(f.l2w), fp(dout), mp(dn), mp(part_dn), mp(sc_n), mp(sc_h), mp(sc_da), mp(sc_dz), B, C, R, V,
                                 f.rs, (unsigned long long)f.site, (float)f.p, s_);
                float *dw1 = grad_ptr(f.l1w), *db1 = grad_ptr(f.l1b), *dw2 = grad_ptr(f.l2w), *db2 = grad_ptr(f.l2b);
                wgrad_submit(s_, f.x.device().index(), [=](void* s) {
                    const void* ptrs[8] = {fp(sc_h), fp(sc_dz), dw2, db2, fp(sc_n), fp(sc_da), dw1, db1};
                    const long dims[8] = {R, C, V, B, C, R, V, B};
                    VX(vx_pw_wgrad_group, ptrs, dims, 2, nullptr, nullptr, 0, s);
                });
            } else
            VX(vx_mlp_bwd_h, (const void*)f.o.data_ptr(), 0, fp(f.stats_o), nullptr, nullptr, fp(f.l1w), fp(f.l1b), fp(f.l2w), fp(dout), dn.data_ptr(), mp(part_dn), nullptr, nullptr,
                           grad_ptr(f.l1w), grad_ptr(f.l1b), grad_ptr(f.l2w), grad_ptr(f.l2b), B, C, R, V, (float)F.in_eps, f.rs, 0ull, 0.0f,
                           (unsigned long long)f.site, (float)f.p, h16, s_);
            const char* yp = (const char*)f.y.data_ptr();
            const void *y1 = yp, *y3 = yp + esz * n1, *y5 = yp + 2 * esz * n1;
            Tensor d_o = at::empty(f.x.sizes(), iopt), part_t = at::empty({3, BC, (long)f.nch, 2}, f.x.options());
            VX(vx_jlc_mid_bwd_h, fp(dout), (const void*)dn.data_ptr(), fp(part_dn), npd, (const void*)f.o.data_ptr(), fp(f.stats_o), y1, y3, y5, fp(f.stats_y), d_o.data_ptr(), mp(part_t), BC, V, h16, s_);
            Tensor g = at::empty({3, B, C, D, H, W}, iopt);
            char* gp = (char*)g.data_ptr();
            void *g1 = gp, *g3 = gp + esz * n1, *g5 = gp + 2 * esz * n1;
            VX(vx_jlc_gk_h, (const void*)d_o.data_ptr(), y1, y3, y5, fp(f.stats_y), fp(part_t), g1, g3, g5, BC, V, h16, s_);
            Tensor dx;
            if (need_x) {
                dx = f.h16 ? at::empty_like(f.x) : dn;           // fp32 storage: dn is dead after vx_jlc_mid_bwd, reuse it (the 16-bit dn cannot hold the block's fp32 input gradient)
                if (f.img.defined() && f.img_cl) VX(vx_jlc_cl_bwd, (const float*)g1, (const float*)g3, (const float*)g5, fp(f.img), fp(d_o), mp(dx), B, C, G, D, H, W, s_);
                else if (f.img.defined()) VX(vx_jlc_tz_bwd_h, (const void*)g1, (const void*)g3, (const void*)g5, fp(f.img), fp(f.w1), (const void*)d_o.data_ptr(), mp(dx), B, C, G, D, H, W, f.tz_pieces, h16, s_);
                else
                VX(vx_jlc_conv_bwd, (const float*)g1, (const float*)g3, (const float*)g5, fp(f.w1), fp(f.w3), fp(f.w5), fp(d_o), mp(dx), B, C, G, D, H, W, s_);
            }
            // weight gradients (the bias gradients behind an InstanceNorm are zero by construction: see below)
            grad_ptr(f.b1); grad_ptr(f.b3); grad_ptr(f.b5);
            const int Cg = C / G;
            {
                // (through wgrad_submit: immediate by default; deferred to the end of this stream in the taped encoder backward -- the closure owns x and g)
                Tensor xk = f.x, gk = g;
                float* dw1 = f.w1.requires_grad() ? grad_ptr(f.w1) : nullptr;
                float* dw3 = f.w3.requires_grad() ? grad_ptr(f.w3) : nullptr;
                float* dw5 = f.w5.requires_grad() ? grad_ptr(f.w5) : nullptr;
                const bool g1fast = F.use_gconv1 && (Cg == 4 || Cg == 8 || Cg == 16) && V % 4 == 0;
                // small volumes: gather-GEMMs in one launch (conv_wgrad.hip) -- where the Toeplitz kernel does not reach (W % 4: the 6^3 / 3^3 levels of the 96^3 configurations)
                // and at <= 64 voxels, where it is the faster one (4^3: 16 vs 27 us; at 8^3 the Toeplitz kernel wins 40 vs 57 us)
                const bool tzok = F.jlc_wg_tz && dw1 && dw3 && dw5 && vx_jlc_wgrad_tz_ok(C, G, D, H, W);
                const bool wga = F.jlc_wg_tz && dw1 && dw3 && dw5 && vx_jlc_wgrad_gather_ok(C, G, D, H, W) == 1 && (!tzok || V <= 64);
                const bool wtz = !wga && tzok;
                const int wg_pieces = f.tz_pieces ? f.tz_pieces : vx_jlc_tz_pieces();          // fixed NOW: the closure may be launched at the end of the encoder backward
                wgrad_submit(s_, f.x.device().index(), [=](void* s) {
                    if (h16) {                    // bf16 storage mode (Toeplitz kernels only: f.h16 implies tz)
                        const char* gh = (const char*)gk.data_ptr();
                        TORCH_CHECK(wtz, "veloxseg_amd.jlc: the 16-bit g_k of the bf16 storage mode need the matrix-pipe weight-gradient kernel");
                        VX(vx_jlc_wgrad_tz_h, fp(xk), (const void*)gh, (const void*)(gh + 2 * n1), (const void*)(gh + 4 * n1), dw1, dw3, dw5, B, C, G, D, H, W, wg_pieces, 1, s);
                        return;
                    }
                    const float* gq = gk.data_ptr<float>();
                    if (wga) {
                        VX(vx_jlc_wgrad_gather, fp(xk), gq, gq + n1, gq + 2 * n1, dw1, dw3, dw5, B, C, G, D, H, W, s);
                        return;
                    }
                    if (wtz) {                    // all three weight gradients in one launch on the matrix pipe (csrc/jlc_mfma.hip)
                        VX(vx_jlc_wgrad_tz_ns, fp(xk), gq, gq + n1, gq + 2 * n1, dw1, dw3, dw5, B, C, G, D, H, W, wg_pieces, s);
                        return;
                    }
                    if (dw1) {
                        if (g1fast) VX(vx_gconv1_bwd_weight, fp(xk), gq, dw1, nullptr, B, C, G, V, s);
                        else VX(vx_conv3d_bwd_weight_tiled, fp(xk), nullptr, 0, gq, dw1, nullptr, B, C, D, H, W, C, 1, 1, 0, G, 1, s);
                    }
                    if (dw3) VX(vx_conv3d_bwd_weight_tiled, fp(xk), nullptr, 0, gq + n1, dw3, nullptr, B, C, D, H, W, C, 3, 1, 1, G, 1, s);
                    if (dw5) VX(vx_conv3d_bwd_weight_tiled, fp(xk), nullptr, 0, gq + 2 * n1, dw5, nullptr, B, C, D, H, W, C, 5, 1, 2, G, 1, s);
                });
                if (!WG.enabled) WG.done.clear();
            }
            st.reset();
            return dx;
        }
        const Tensor& dout = dout_in;
        Tensor do_res, dz, dh, dh2, da, dn, dn2;
        axpy_bwd_impl(st->r, dout, true, do_res, dz, s_);
        if (st->fused) da = pw_gelu_bwd_impl(st->c2, st->g, dz, s_);
        else {
            conv_bwd_impl(st->c2, dz, true, dh, dh2, s_);
            da = gelu_bwd_impl(st->g, dh, s_);
        }
        conv_bwd_impl(st->c1, da, true, dn, dn2, s_);
        Tensor d_o;                                            // = do_res + InstanceNorm backward of the channel stage's input
        if (F.fuse_bwd_add) d_o = in_bwd_impl(st->in2, dn, {true}, s_, nullptr, 1, do_res)[0];
        else {
            Tensor do2 = in_bwd_impl(st->in2, dn, {true}, s_)[0];
            d_o = sum3(do_res, do2, Tensor(), s_);
        }
        const int n = (int)st->convs.size();
        // The spatial convs feed an InstanceNorm directly: its backward removes the per-(b, c) mean, so the sum of g[k] over every row -- the bias
        // gradient -- is zero up to round-off (the reference's value is ~1e-8 noise).  It is not computed: no reduction launch on the long rows,
        // no per-row atomics on the short ones; the bias entries of the flat gradient stay at the zero they were cleared to.
        auto g = in_bwd_impl(st->in1, d_o, std::vector<bool>(n, true), s_);
        // dx = d_o + sum_k conv_k^T(g_k): d_o is this function's own temporary (its last reader was in_bwd above), so the three input gradients
        // are accumulated into it in place by the kernels' `accumulate` mode -- no extra add launches
        for (int k = 0; k < n; ++k) {
            Tensor t1, t2;
            if (F.skip_in_bias) grad_ptr(st->convs[k].b);         // the (zero) gradient tensor still exists for optimizers that walk p.grad
            conv_bwd_impl(st->convs[k], g[k], true, t1, t2, s_, d_o, /*skip_bias=*/F.skip_in_bias);
        }
        st.reset();
        if (!need_x) return Tensor();
        return d_o;
    }

static std::pair<Tensor, std::shared_ptr<FFNState>> ffn_fwd_f(const Tensor& y, const Tensor& gamma, const Tensor& beta, const Tensor& w1, const Tensor& b1, const Tensor& w2, const Tensor& b2, double p,
                        int64_t site1, int64_t site2, int64_t rs, int64_t stream) {
        auto st = std::make_shared<FFNState>();
        void* s_ = sp(stream);
        if (F.fuse_blocks && y.dim() == 5) {
            const int B = y.size(0), C = y.size(1), R = w1.size(0);
            const long V = y.numel() / ((long)B * C);
            if (vx_mlp_supported(C, R, V)) {
                check_in(y, "ffn");
                st->blk = true;
                st->y = contig(y); st->gamma = gamma; st->beta = beta; st->w1 = w1; st->b1 = b1; st->w2 = w2; st->b2 = b2;
                st->p = p; st->site1 = site1; st->site2 = site2; st->rs = p > 0 ? sp(rs) : nullptr;
                Tensor out = at::empty_like(st->y);
                VX(vx_mlp_fwd, fp(st->y), 1, nullptr, 0, nullptr, fp(gamma), fp(beta), fp(w1), fp(b1), fp(w2), fp(b2), mp(out), B, C, R, V, (float)F.ln_eps, st->rs,
                               (unsigned long long)site1, (float)p, (unsigned long long)site2, (float)p, s_);
                return {out, st};
            }
        }
        Tensor n = ln_fwd_impl(st->ln, y, gamma, beta, s_);
        st->fused = F.fuse_gelu && pw_gelu_fusable(n, w1) && w2.size(1) % 4 == 0;
        Tensor h;
        if (st->fused) h = pw_gelu_fwd_impl(st->c1, st->g, n, w1, b1, p, site1, sp(rs), s_);
        else {
            Tensor a = conv_fwd_impl(st->c1, n, Tensor(), w1, b1, 1, 1, 0, 1, 1, true, s_);
            h = gelu_fwd_impl(st->g, a, p, site1, sp(rs), s_);
        }
        Tensor out;
        if (F.fuse_res && pw_gelu_fusable(h, w2)) out = pw_res_fwd_impl(st->c2, st->r, h, w2, b2, y, 1.0, p, site2, sp(rs), s_);
        else {
            Tensor z = conv_fwd_impl(st->c2, h, Tensor(), w2, b2, 1, 1, 0, 1, 1, true, s_);
            out = axpy_fwd_impl(st->r, y, z, 1.0, p, site2, sp(rs), s_);
        }
        return {out, st};
    }

static Tensor ffn_bwd_f(std::shared_ptr<FFNState> st, const Tensor& dout_in, int64_t stream) {
        void* s_ = sp(stream);
        if (st->blk) {
            Tensor dout = contig(dout_in);
            const int B = st->y.size(0), C = st->y.size(1), R = st->w1.size(0);
            const long V = st->y.numel() / ((long)B * C);
            Tensor dy = at::empty_like(st->y);
            VX(vx_mlp_bwd, fp(st->y), 1, nullptr, fp(st->gamma), fp(st->beta), fp(st->w1), fp(st->b1), fp(st->w2), fp(dout), mp(dy), nullptr, grad_ptr(st->gamma),
                           grad_ptr(st->beta), grad_ptr(st->w1), grad_ptr(st->b1), grad_ptr(st->w2), grad_ptr(st->b2), B, C, R, V, (float)F.ln_eps, st->rs,
                           (unsigned long long)st->site1, (float)st->p, (unsigned long long)st->site2, (float)st->p, s_);
            return dy;
        }
        const Tensor& dout = dout_in;
        Tensor dy_res, dz, dh, t2, dn, t3;
        axpy_bwd_impl(st->r, dout, true, dy_res, dz, s_);
        Tensor da;
        if (st->fused) da = pw_gelu_bwd_impl(st->c2, st->g, dz, s_);
        else {
            conv_bwd_impl(st->c2, dz, true, dh, t2, s_);
            da = gelu_bwd_impl(st->g, dh, s_);
        }
        conv_bwd_impl(st->c1, da, true, dn, t3, s_);
        if (F.fuse_bwd_add) return ln_bwd_impl(st->ln, dn, s_, dy_res);      // dy_res + LayerNorm backward in one store
        Tensor dy_ln = ln_bwd_impl(st->ln, dn, s_);
        return sum3(dy_res, dy_ln, Tensor(), s_);
    }

// JLC block / FFN tail as C++ autograd nodes (the parameters are passed so that the node exists even when only they require a gradient;
// their gradients are accumulated into .grad by the kernels, as everywhere)
struct JLCFn : public torch::autograd::Function<JLCFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& w0, const OptT& w1, const OptT& w2, const Tensor& b0, const OptT& b1, const OptT& b2, int64_t G,
                          const Tensor& l1w, const Tensor& l1b, const Tensor& l2w, const Tensor& l2b, double p, int64_t site, int64_t rs) {
        std::vector<Tensor> ws{w0}, bs{b0};
        if (w1.has_value() && w1->defined()) { ws.push_back(*w1); bs.push_back(b1.value()); }
        if (w2.has_value() && w2->defined()) { ws.push_back(*w2); bs.push_back(b2.value()); }
        auto r = jlc_fwd_f(x, ws, bs, (int)G, l1w, l1b, l2w, l2b, p, site, rs, (int64_t)cur_stream(x));
        put_state<std::shared_ptr<JLCState>>(ctx)->s = r.second;
        ctx->saved_data["rng"] = take_rng_keep();
        return r.first;
    }
    static variable_list backward(AutogradContext* ctx, variable_list g) {
        auto st = get_state<std::shared_ptr<JLCState>>(ctx);
        Tensor dx = jlc_bwd_f(st, g[0], ctx->needs_input_grad(0), (int64_t)cur_stream(g[0]));
        ctx->saved_data.clear();
        variable_list out(15);
        out[0] = dx;
        return out;
    }
};
struct FFNFn : public torch::autograd::Function<FFNFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& y, const Tensor& gamma, const Tensor& beta, const Tensor& w1, const Tensor& b1, const Tensor& w2, const Tensor& b2,
                          double p, int64_t site1, int64_t site2, int64_t rs) {
        auto r = ffn_fwd_f(y, gamma, beta, w1, b1, w2, b2, p, site1, site2, rs, (int64_t)cur_stream(y));
        put_state<std::shared_ptr<FFNState>>(ctx)->s = r.second;
        ctx->saved_data["rng"] = take_rng_keep();
        return r.first;
    }
    static variable_list backward(AutogradContext* ctx, variable_list g) {
        auto st = get_state<std::shared_ptr<FFNState>>(ctx);
        Tensor dy = ffn_bwd_f(st, g[0], (int64_t)cur_stream(g[0]));
        ctx->saved_data.clear();
        variable_list out(11);
        out[0] = dy;
        return out;
    }
};


// The VeloxSeg loss (utils/loss.py:52-66: sum_h w_h (CE + Dice)(logits_h) + w_rc MSE(rcs, x) + w_f / M sum_m MSE(G_seg, G_m)) as a C++ autograd node: the same C-ABI
// sequence as functional._VeloxLossFn (deep-supervision heads on coarser grids are interpolated inside the kernels of loss_ds.hip), no interpreter, no Python state.
// outputs = nh logits [+ rcs, G_seg, G_rc x M]
struct LossState {
    std::vector<Tensor> logits, tail;          // tail: rcs, sr, G_seg, G_m...
    Tensor labels, coef;
    std::vector<int> dims;                     // fused deep supervision: (d, h, w) of heads 1..
    int nh = 0, M = 0, B = 0, C = 0, D = 0, H = 0, W = 0, lab_kind = 0;
    long V = 0;
    bool ds = false, has_tail = false;
};
inline int lab_kind_of(const Tensor& labels) {
    if (labels.scalar_type() == at::kLong) return 0;
    if (labels.scalar_type() == at::kInt) return 1;
    if (labels.scalar_type() == at::kByte) return 2;
    TORCH_CHECK(false, "veloxseg::seg_loss: labels must be int64 / int32 / uint8, got ", labels.scalar_type());
}
// the (at most four) head weights as a device tensor, cached per (values, device) as functional._head_weights does: a blocking pageable H2D copy per call would stall the
// stream and is illegal inside a stream capture.  The first call of a new weight set must happen outside a capture (TrainEngine's warm-up pass does).
inline Tensor head_weights_on(const std::vector<float>& w, const c10::Device& dev) {
    static std::map<std::pair<std::vector<float>, int>, Tensor> cache;
    auto key = std::make_pair(w, (int)dev.index());
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    Tensor t = at::tensor(w, at::TensorOptions().dtype(at::kFloat)).to(dev, /*non_blocking=*/false);
    cache.emplace(key, t);
    return t;
}
struct LossFn : public torch::autograd::Function<LossFn> {
    static Tensor forward(AutogradContext* ctx, at::TensorList outs, const Tensor& labels_in, const OptT& sr_in, std::vector<double> head_w, double w_rc, double w_f, int64_t num_modal) {
        auto h = put_state<LossState>(ctx);
        LossState& st = h->s;
        const int M = (int)num_modal;
        const int nh = (int)outs.size() - (M > 0 ? 2 + M : 0);
        TORCH_CHECK(nh >= 1 && nh <= 4 && (int)head_w.size() == nh, "veloxseg::seg_loss: 1..4 deep-supervision heads with one weight each (got ", nh, " heads, ", head_w.size(), " weights)");
        check_in(outs[0], "seg_loss");
        for (int i = 0; i < nh; ++i) st.logits.push_back(contig(outs[i]));
        const Tensor& l0 = st.logits[0];
        TORCH_CHECK(l0.dim() == 5, "veloxseg::seg_loss: logits are (B, C, D, H, W)");
        st.nh = nh; st.M = M; st.B = (int)l0.size(0); st.C = (int)l0.size(1); st.D = (int)l0.size(2); st.H = (int)l0.size(3); st.W = (int)l0.size(4);
        st.V = (long)st.D * st.H * st.W;
        st.labels = contig(labels_in);
        st.lab_kind = lab_kind_of(st.labels);
        TORCH_CHECK(st.labels.is_cuda() && st.labels.numel() == (long)st.B * st.V, "veloxseg::seg_loss: labels must be a CUDA tensor of B x D x H x W elements");
        void* s_ = cur_stream(l0);
        const int B = st.B, C = st.C;
        auto fopt = l0.options();
        Tensor hw = head_weights_on(std::vector<float>(head_w.begin(), head_w.end()), l0.device());
        Tensor seg_acc = at::empty({(long)nh * (1 + (long)B * C * 3)}, fopt.dtype(at::kDouble));
        const float* lp[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int i = 0; i < nh; ++i) lp[i] = fp(st.logits[i]);
        for (int i = 1; i < nh; ++i) if (st.logits[i].sizes().slice(2) != l0.sizes().slice(2)) st.ds = true;
        if (st.ds) {
            TORCH_CHECK(vx_seg_loss_ds_ok(C, st.D, st.H, st.W) == 1, "veloxseg::seg_loss: deep-supervision heads on coarser grids need C in 2..4 and W % 4 == 0 with W / 4 dividing 64; up-sample them first");
            for (int i = 1; i < nh; ++i) for (int k = 2; k < 5; ++k) st.dims.push_back((int)st.logits[i].size(k));
            if (st.dims.empty()) st.dims.assign(3, 1);
            VX(vx_seg_loss_ds_fwd, lp[0], lp[1], lp[2], lp[3], st.dims.data(), nh, (const void*)st.labels.data_ptr(), st.lab_kind, seg_acc.data_ptr<double>(), B, C, st.D, st.H, st.W, s_);
        } else
            VX(vx_seg_loss_fwd, lp[0], lp[1], lp[2], lp[3], nh, (const void*)st.labels.data_ptr(), st.lab_kind, seg_acc.data_ptr<double>(), B, C, st.V, s_);
        st.has_tail = M > 0;
        Tensor rc_acc;
        if (st.has_tail) {
            TORCH_CHECK(sr_in.has_value() && sr_in->defined(), "veloxseg::seg_loss: sr_labels (the network input) is needed for the reconstruction term");
            Tensor rcs = contig(outs[nh]), sr = contig(*sr_in);
            TORCH_CHECK(rcs.sizes() == sr.sizes(), "veloxseg::seg_loss: reconstructions and sr_labels differ in shape");
            st.tail.push_back(rcs); st.tail.push_back(sr);
            for (int i = 0; i < 1 + M; ++i) st.tail.push_back(contig(outs[nh + 1 + i]));
            rc_acc = at::empty({1}, fopt.dtype(at::kDouble));
            VX(vx_sqdiff_sum, fp(rcs), fp(sr), (long)rcs.numel(), rc_acc.data_ptr<double>(), s_);
        }
        st.coef = at::empty({(long)nh * (1 + (long)B * C * 2) + 2}, fopt);
        Tensor loss = at::empty({}, fopt);                    // 0-dim (NOT loss.view({}): the empty braces pick view(ScalarType) with dtype Byte)
        const float* gp[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int m = 0; m < M && m < 4; ++m) gp[m] = fp(st.tail[3 + m]);
        VX(vx_loss_finalize, seg_acc.data_ptr<double>(), nh, B, C, st.V, fp(hw), st.has_tail ? rc_acc.data_ptr<double>() : (const double*)nullptr, st.has_tail ? (long)st.tail[0].numel() : 1L, (float)w_rc,
           st.has_tail ? fp(st.tail[2]) : (const float*)nullptr, gp[0], gp[1], gp[2], gp[3], st.has_tail ? M : 0, st.has_tail ? (int)st.tail[2].size(1) : 0, (float)w_f, mp(loss), mp(st.coef), s_);
        return loss;
    }
    static variable_list backward(AutogradContext* ctx, variable_list g) {
        LossState& st = get_state<LossState>(ctx);
        const int nh = st.nh, M = st.M, B = st.B, C = st.C;
        void* s_ = cur_stream(g[0]);
        Tensor go = contig(g[0].reshape({1}).to(at::kFloat));
        const int nout = nh + (st.has_tail ? 2 + M : 0);
        variable_list out(nout + 6);                          // the list's tensors first (argument order), then labels, sr, head_w, w_rc, w_f, num_modal
        const long stride = 1 + (long)B * C * 2;
        std::vector<Tensor> grads;
        const float* lp[4] = {nullptr, nullptr, nullptr, nullptr};
        float* dp[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int i = 0; i < nh; ++i) { grads.push_back(at::empty_like(st.logits[i])); lp[i] = fp(st.logits[i]); dp[i] = grads[i].data_ptr<float>(); }
        if (st.ds) {
            const int nws = vx_seg_loss_ds_ws_floats(st.dims.data(), nh, B, C, st.D);
            TORCH_CHECK(nws >= 0, "vx_seg_loss_ds_ws_floats failed");
            Tensor ws = at::empty({(long)std::max(nws, 1)}, st.coef.options());
            VX(vx_seg_loss_ds_bwd, lp[0], lp[1], lp[2], lp[3], st.dims.data(), nh, (const void*)st.labels.data_ptr(), st.lab_kind, fp(st.coef), (int)stride, fp(go), dp[0], dp[1], dp[2], dp[3], mp(ws),
               B, C, st.D, st.H, st.W, s_);
        } else
            VX(vx_seg_loss_bwd4, lp[0], lp[1], lp[2], lp[3], nh, (const void*)st.labels.data_ptr(), st.lab_kind, fp(st.coef), (int)stride, fp(go), dp[0], dp[1], dp[2], dp[3], B, C, st.V, s_);
        for (int i = 0; i < nh; ++i) out[i] = grads[i];
        if (st.has_tail) {
            const float* misc = fp(st.coef) + (long)nh * stride;
            Tensor drc = at::empty_like(st.tail[0]);
            VX(vx_mse_bwd, fp(st.tail[0]), fp(st.tail[1]), misc, fp(go), mp(drc), (long)drc.numel(), s_);
            out[nh] = drc;
            Tensor dgs = at::empty_like(st.tail[2]);
            std::vector<Tensor> dgm;
            const float* gp[4] = {nullptr, nullptr, nullptr, nullptr};
            float* dgp[4] = {nullptr, nullptr, nullptr, nullptr};
            for (int m = 0; m < M && m < 4; ++m) { dgm.push_back(at::empty_like(st.tail[3 + m])); gp[m] = fp(st.tail[3 + m]); dgp[m] = dgm[m].data_ptr<float>(); }
            VX(vx_gram_mse_bwd, fp(st.tail[2]), gp[0], gp[1], gp[2], gp[3], M, misc + 1, fp(go), mp(dgs), dgp[0], dgp[1], dgp[2], dgp[3], (long)dgs.numel(), s_);
            out[nh + 1] = dgs;
            for (int m = 0; m < M; ++m) out[nh + 2 + m] = dgm[m];
        }
        ctx->saved_data.clear();
        return out;
    }
};

}  // namespace

// =================================================================================================================== python surface
// ---------------------------------------------------------------------------------------------------------------------------------------------
// Dispatcher registration (north_star: "re-registered as custom ops backed by hand-written HIP kernels").  Schemas are defined HERE (veloxseg_amd/ops.py lists them and adds
// the operators whose bodies need Python-side state: dropout sites, the PWA plan, the loss).  Keys:
//   Autograd  -> the operator's torch::autograd::Function (forward + backward launch the kernels of include/veloxseg_hip.h)
//   CUDA      -> the same forward without a graph (inference mode reaches this key directly)
//   Meta      -> shapes only (FakeTensor / torch.export / torch.compile tracing)
//   CPU       -> raises: the hot path has no CPU kernel, and the dispatcher's own "no kernel for backend CPU" message would not say why
namespace {
using torch::autograd::AutogradContext;
[[noreturn]] void op_no_cpu(const char* what) {
    TORCH_CHECK(false, "veloxseg::", what, ": the VeloxSeg hot path runs only on an MI355X (HIP kernels); there is deliberately no CPU kernel behind this operator");
}
int64_t conv_out(int64_t n, int64_t K, int64_t S, int64_t P) { return (n + 2 * P - K) / S + 1; }
Tensor op_conv3d_meta(const Tensor&, const Tensor&, const OptT&, int64_t, int64_t, int64_t, int64_t);
Tensor op_upconv_meta(const Tensor&, const Tensor&, const Tensor&);
Tensor op_in_sum_meta(at::TensorList, bool, const OptT&);
Tensor op_ln_meta(const Tensor&, const Tensor&, const Tensor&);
Tensor op_s2d_meta(const Tensor&);
Tensor op_up_meta(const Tensor&, at::IntArrayRef);
Tensor op_gram_meta(const Tensor&);

Tensor op_conv3d(const Tensor& x, const Tensor& w, const OptT& b, int64_t stride, int64_t padding, int64_t groups, int64_t ps) {
    if (x.is_meta()) return op_conv3d_meta(x, w, b, stride, padding, groups, ps);     // (the Autograd key also covers meta and CPU tensors)
    if (!x.is_cuda()) op_no_cpu("conv3d");           // fail before anything touches HIP
    return ConvFn::apply(x, OptT(), w, b, w.size(2), stride, padding, groups, ps);
}
Tensor op_conv3d_meta(const Tensor& x, const Tensor& w, const OptT&, int64_t S, int64_t P, int64_t, int64_t ps) {
    const int64_t K = w.size(2), r = ps * ps * ps;
    TORCH_CHECK(x.dim() == 5 && w.dim() == 5 && w.size(0) % r == 0, "veloxseg::conv3d: bad shapes");
    return at::empty({x.size(0), w.size(0) / r, conv_out(x.size(2), K, S, P) * ps, conv_out(x.size(3), K, S, P) * ps, conv_out(x.size(4), K, S, P) * ps}, x.options());
}
Tensor op_conv3d_cpu(const Tensor&, const Tensor&, const OptT&, int64_t, int64_t, int64_t, int64_t) { op_no_cpu("conv3d"); }

Tensor op_upconv(const Tensor& x, const Tensor& w, const Tensor& b) { if (x.is_meta()) return op_upconv_meta(x, w, b); if (!x.is_cuda()) op_no_cpu("conv_transpose_k2s2"); return UpconvFn::apply(x, w, OptT(b), false); }
Tensor op_upconv_meta(const Tensor& x, const Tensor& w, const Tensor&) { return at::empty({x.size(0), w.size(1), 2 * x.size(2), 2 * x.size(3), 2 * x.size(4)}, x.options()); }
Tensor op_upconv_cpu(const Tensor&, const Tensor&, const Tensor&) { op_no_cpu("conv_transpose_k2s2"); }

Tensor op_in_sum(at::TensorList ys, bool act, const OptT& res) {
    TORCH_CHECK(ys.size() >= 1 && ys.size() <= 3, "veloxseg::instance_norm_sum: 1..3 inputs");
    if (ys[0].is_meta()) return op_in_sum_meta(ys, act, res);
    if (!ys[0].is_cuda()) op_no_cpu("instance_norm_sum");
    return InstNormFn::apply(res, act, ys[0], ys.size() > 1 ? OptT(ys[1]) : OptT(), ys.size() > 2 ? OptT(ys[2]) : OptT());
}
Tensor op_in_sum_meta(at::TensorList ys, bool, const OptT&) { return at::empty_like(ys[0]); }
Tensor op_in_sum_cpu(at::TensorList, bool, const OptT&) { op_no_cpu("instance_norm_sum"); }

Tensor op_ln(const Tensor& x, const Tensor& g, const Tensor& bt) { if (x.is_meta()) return op_ln_meta(x, g, bt); if (!x.is_cuda()) op_no_cpu("layer_norm_cf"); return LayerNormFn::apply(x, g, bt); }
Tensor op_ln_meta(const Tensor& x, const Tensor&, const Tensor&) { return at::empty_like(x); }
Tensor op_ln_cpu(const Tensor&, const Tensor&, const Tensor&) { op_no_cpu("layer_norm_cf"); }

Tensor op_s2d(const Tensor& x) { if (x.is_meta()) return op_s2d_meta(x); if (!x.is_cuda()) op_no_cpu("space_to_depth2"); return S2DFn::apply(x); }
Tensor op_s2d_meta(const Tensor& x) { return at::empty({x.size(0), 8 * x.size(1), x.size(2) / 2, x.size(3) / 2, x.size(4) / 2}, x.options()); }
Tensor op_s2d_cpu(const Tensor&) { op_no_cpu("space_to_depth2"); }

Tensor op_up(const Tensor& x, at::IntArrayRef size) {
    TORCH_CHECK(size.size() == 3, "veloxseg::upsample_trilinear: size = (D, H, W)");
    if (x.is_meta()) return op_up_meta(x, size);
    if (!x.is_cuda()) op_no_cpu("upsample_trilinear");
    return UpsampleFn::apply(x, size[0], size[1], size[2]);
}
Tensor op_up_meta(const Tensor& x, at::IntArrayRef size) { return at::empty({x.size(0), x.size(1), size[0], size[1], size[2]}, x.options()); }
Tensor op_up_cpu(const Tensor&, at::IntArrayRef) { op_no_cpu("upsample_trilinear"); }

Tensor op_gram(const Tensor& x) { if (x.is_meta()) return op_gram_meta(x); if (!x.is_cuda()) op_no_cpu("gram"); return GramFn::apply(x); }
Tensor op_gram_meta(const Tensor& x) { return at::empty({x.size(0), x.size(1), x.size(1)}, x.options()); }
Tensor op_gram_cpu(const Tensor&) { op_no_cpu("gram"); }
}  // namespace

// ---- the operators north_star names: PWA attention (PWA.py:329), the JLC block (conv_blocks.py:72), the FFN tail (attention_utils.py:45-71), the loss (utils/loss.py:50-66).
// Everything they need arrives as an argument -- the window plan as integer lists, the dropout site and the {seed, step} RNG-state tensor (veloxseg_amd.functional.rng_state)
// -- so no Python state stands behind any key.
namespace {
const void* rs_of(const OptT& rng_state, double p, const char* what) {
    if (p <= 0.0) return nullptr;
    TORCH_CHECK(rng_state.has_value() && rng_state->defined(), "veloxseg::", what, ": dropout p > 0 needs rng_state = the {seed, step} int64 device tensor (veloxseg_amd.functional.rng_state(device))");
    TORCH_CHECK(rng_state->is_cuda() && rng_state->scalar_type() == at::kLong && rng_state->numel() >= 2 && rng_state->is_contiguous(), "veloxseg::", what, ": rng_state must be a contiguous CUDA int64 tensor {seed, step}");
    RNG_KEEP = *rng_state;          // the node's state takes a reference: the backward reads {seed, step} through this tensor's address
    return rng_state->data_ptr();
}
VxPwaPlan plan_of(at::IntArrayRef grid, at::IntArrayRef n, int64_t heads, at::IntArrayRef small, at::IntArrayRef nwin) {
    TORCH_CHECK(grid.size() == 3 && n.size() == 3 && small.size() % 3 == 0 && small.size() == nwin.size() && small.size() >= 3 && small.size() <= 12,
                "veloxseg::pwa_attention: grid / n are 3 integers, small / nwin 3 per window scale (1..4 scales)");
    VxPwaPlan pl = {};
    const int nb = (int)small.size() / 3;
    for (int k = 0; k < 3; ++k) { pl.grid[k] = (int)grid[k]; pl.n[k] = (int)n[k]; }
    pl.heads = (int)heads; pl.nb = nb;
    int off = 0;
    for (int i = 0; i < nb; ++i) {
        for (int k = 0; k < 3; ++k) { pl.small[i][k] = (int)small[3 * i + k]; pl.nwin[i][k] = (int)nwin[3 * i + k]; }
        pl.woff[i] = off;
        off += (int)(nwin[3 * i] * nwin[3 * i + 1] * nwin[3 * i + 2]);
    }
    pl.Ntot = off;
    pl.l = (int)(n[0] * n[1] * n[2]);
    return pl;
}
std::vector<Tensor> op_pwa_meta(const Tensor&, at::TensorList qkv, at::IntArrayRef grid, at::IntArrayRef, int64_t heads, at::IntArrayRef small, at::IntArrayRef, int64_t, int64_t cv, double, int64_t, const OptT&) {
    TORCH_CHECK(qkv.size() >= 3 && qkv.size() % 3 == 0, "veloxseg::pwa_attention: qkv = (q, k, v) per modality");
    std::vector<Tensor> out;
    for (size_t m = 0; m < qkv.size() / 3; ++m) out.push_back(at::empty({qkv[0].size(0), (int64_t)(small.size() / 3) * heads * cv, grid[0], grid[1], grid[2]}, qkv[0].options()));
    return out;
}
std::vector<Tensor> op_pwa(const Tensor& table, at::TensorList qkv, at::IntArrayRef grid, at::IntArrayRef n, int64_t heads, at::IntArrayRef small, at::IntArrayRef nwin, int64_t cq, int64_t cv,
                           double p_attn, int64_t site, const OptT& rng_state) {
    TORCH_CHECK(qkv.size() >= 3 && qkv.size() % 3 == 0, "veloxseg::pwa_attention: qkv = (q, k, v) per modality");
    if (qkv[0].is_meta()) return op_pwa_meta(table, qkv, grid, n, heads, small, nwin, cq, cv, p_attn, site, rng_state);
    if (!qkv[0].is_cuda()) op_no_cpu("pwa_attention");
    VxPwaPlan pl = plan_of(grid, n, heads, small, nwin);
    return PwaCoreFn::apply(table, (int64_t)reinterpret_cast<intptr_t>(&pl), cq, cv, p_attn, site, (int64_t)reinterpret_cast<intptr_t>(rs_of(rng_state, p_attn, "pwa_attention")), qkv);
}
std::vector<Tensor> op_pwa_cpu(const Tensor&, at::TensorList, at::IntArrayRef, at::IntArrayRef, int64_t, at::IntArrayRef, at::IntArrayRef, int64_t, int64_t, double, int64_t, const OptT&) { op_no_cpu("pwa_attention"); }

Tensor op_jlc_meta(const Tensor& x, at::TensorList, at::TensorList, int64_t, const Tensor&, const Tensor&, const Tensor&, const Tensor&, double, int64_t, const OptT&) { return at::empty_like(x); }
Tensor op_jlc(const Tensor& x, at::TensorList ws, at::TensorList bs, int64_t groups, const Tensor& l1w, const Tensor& l1b, const Tensor& l2w, const Tensor& l2b, double p, int64_t site,
              const OptT& rng_state) {
    TORCH_CHECK(ws.size() >= 1 && ws.size() <= 3 && bs.size() == ws.size(), "veloxseg::jlc_block: 1..3 grouped convolutions with one bias each");
    if (x.is_meta()) return op_jlc_meta(x, ws, bs, groups, l1w, l1b, l2w, l2b, p, site, rng_state);
    if (!x.is_cuda()) op_no_cpu("jlc_block");
    return JLCFn::apply(x, ws[0], ws.size() > 1 ? OptT(ws[1]) : OptT(), ws.size() > 2 ? OptT(ws[2]) : OptT(), bs[0], bs.size() > 1 ? OptT(bs[1]) : OptT(), bs.size() > 2 ? OptT(bs[2]) : OptT(), groups,
                        l1w, l1b, l2w, l2b, p, site, (int64_t)reinterpret_cast<intptr_t>(rs_of(rng_state, p, "jlc_block")));
}
Tensor op_jlc_cpu(const Tensor&, at::TensorList, at::TensorList, int64_t, const Tensor&, const Tensor&, const Tensor&, const Tensor&, double, int64_t, const OptT&) { op_no_cpu("jlc_block"); }

Tensor op_ffn_meta(const Tensor& y, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, double, int64_t, int64_t, const OptT&) { return at::empty_like(y); }
Tensor op_ffn(const Tensor& y, const Tensor& gamma, const Tensor& beta, const Tensor& w1, const Tensor& b1, const Tensor& w2, const Tensor& b2, double p, int64_t site1, int64_t site2,
              const OptT& rng_state) {
    if (y.is_meta()) return op_ffn_meta(y, gamma, beta, w1, b1, w2, b2, p, site1, site2, rng_state);
    if (!y.is_cuda()) op_no_cpu("ffn_tail");
    return FFNFn::apply(y, gamma, beta, w1, b1, w2, b2, p, site1, site2, (int64_t)reinterpret_cast<intptr_t>(rs_of(rng_state, p, "ffn_tail")));
}
Tensor op_ffn_cpu(const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, double, int64_t, int64_t, const OptT&) { op_no_cpu("ffn_tail"); }

Tensor op_loss_meta(at::TensorList outs, const Tensor&, const OptT&, at::ArrayRef<double>, double, double, int64_t) { return at::empty({}, outs[0].options()); }
Tensor op_loss(at::TensorList outs, const Tensor& labels, const OptT& sr, at::ArrayRef<double> head_w, double w_rc, double w_f, int64_t num_modal) {
    TORCH_CHECK(outs.size() >= 1, "veloxseg::seg_loss: no outputs");
    if (outs[0].is_meta()) return op_loss_meta(outs, labels, sr, head_w, w_rc, w_f, num_modal);
    if (!outs[0].is_cuda()) op_no_cpu("seg_loss");
    return LossFn::apply(outs, labels, sr, head_w.vec(), w_rc, w_f, num_modal);
}
Tensor op_loss_cpu(at::TensorList, const Tensor&, const OptT&, at::ArrayRef<double>, double, double, int64_t) { op_no_cpu("seg_loss"); }
}  // namespace

TORCH_LIBRARY(veloxseg, m) {
    m.def("conv3d(Tensor x, Tensor w, Tensor? b, int stride, int padding, int groups, int pixel_shuffle) -> Tensor");
    m.def("conv_transpose_k2s2(Tensor x, Tensor w, Tensor b) -> Tensor");
    m.def("instance_norm_sum(Tensor[] ys, bool act, Tensor? res) -> Tensor");
    m.def("layer_norm_cf(Tensor x, Tensor gamma, Tensor beta) -> Tensor");
    m.def("space_to_depth2(Tensor x) -> Tensor");
    m.def("upsample_trilinear(Tensor x, int[] size) -> Tensor");
    m.def("gram(Tensor x) -> Tensor");
    m.def("pwa_attention(Tensor table, Tensor[] qkv, int[] grid, int[] n, int heads, int[] small, int[] nwin, int cq, int cv, float p_attn, int site, Tensor? rng_state=None) -> Tensor[]");
    m.def("jlc_block(Tensor x, Tensor[] ws, Tensor[] bs, int groups, Tensor l1w, Tensor l1b, Tensor l2w, Tensor l2b, float p, int site, Tensor? rng_state=None) -> Tensor");
    m.def("ffn_tail(Tensor y, Tensor gamma, Tensor beta, Tensor w1, Tensor b1, Tensor w2, Tensor b2, float p, int site1, int site2, Tensor? rng_state=None) -> Tensor");
    m.def("seg_loss(Tensor[] outputs, Tensor labels, Tensor? sr_labels, float[] head_weights, float w_rc, float w_f, int num_modal) -> Tensor");
}
#define VX_OP_IMPLS(KEY, SUF)                                 \
    TORCH_LIBRARY_IMPL(veloxseg, KEY, m) {                    \
        m.impl("conv3d", op_conv3d##SUF);                     \
        m.impl("conv_transpose_k2s2", op_upconv##SUF);        \
        m.impl("instance_norm_sum", op_in_sum##SUF);          \
        m.impl("layer_norm_cf", op_ln##SUF);                  \
        m.impl("space_to_depth2", op_s2d##SUF);               \
        m.impl("upsample_trilinear", op_up##SUF);             \
        m.impl("gram", op_gram##SUF);                         \
        m.impl("pwa_attention", op_pwa##SUF);                 \
        m.impl("jlc_block", op_jlc##SUF);                     \
        m.impl("ffn_tail", op_ffn##SUF);                      \
        m.impl("seg_loss", op_loss##SUF);                     \
    }
VX_OP_IMPLS(Autograd, )
VX_OP_IMPLS(CUDA, )
VX_OP_IMPLS(Meta, _meta)
VX_OP_IMPLS(CPU, _cpu)
#undef VX_OP_IMPLS

PYBIND11_MODULE(_vxops, m) {
    m.doc() = "C++ operator bodies of veloxseg_amd.functional (same C-ABI calls, no interpreter in between)";
    py::class_<ConvState, std::shared_ptr<ConvState>>(m, "ConvState");
    py::class_<INState, std::shared_ptr<INState>>(m, "INState");
    py::class_<LNState, std::shared_ptr<LNState>>(m, "LNState");
    py::class_<GeluState, std::shared_ptr<GeluState>>(m, "GeluState");
    py::class_<AxpyState, std::shared_ptr<AxpyState>>(m, "AxpyState");
    py::class_<JLCState, std::shared_ptr<JLCState>>(m, "JLCState");
    py::class_<FFNState, std::shared_ptr<FFNState>>(m, "FFNState");

    m.def("set_flags", [](bool s1, bool expand_mfma, bool gconv1, bool wgrad_ws, bool patchify, bool in_row, int64_t pw_mfma_max_v, int64_t in_row_max, double in_eps,
                          double ln_eps) {
        F.use_s1 = s1; F.use_expand_mfma = expand_mfma; F.use_gconv1 = gconv1; F.use_wgrad_ws = wgrad_ws; F.use_patchify = patchify; F.use_in_row = in_row;
        F.pw_mfma_max_v = pw_mfma_max_v; F.in_row_max = in_row_max; F.in_eps = in_eps; F.ln_eps = ln_eps;
    });

    // C++ autograd nodes: return tensors that already carry their grad_fn
    m.def("conv", [](const Tensor& x, const OptT& x2, const Tensor& w, const OptT& b, int64_t K, int64_t S, int64_t P, int64_t G, int64_t ps) {
        return ConvFn::apply(x, x2, w, b, K, S, P, G, ps);
    });
    // the same node; a patch-expand layer in the bf16 storage mode returns a bfloat16 tensor (any other convolution: fp32 as always)
    m.def("conv_h", [](const Tensor& x, const Tensor& w, const OptT& b, int64_t K, int64_t S, int64_t P, int64_t G, int64_t ps) {
        CONV_OUT_H16 = true;
        auto y = ConvFn::apply(x, OptT(), w, b, K, S, P, G, ps);
        CONV_OUT_H16 = false;
        return y;
    });
    m.def("instnorm", [](const OptT& res, bool act, const Tensor& y0, const OptT& y1, const OptT& y2) { return InstNormFn::apply(res, act, y0, y1, y2); });
    m.def("pw_res", [](const Tensor& x, const Tensor& w, const OptT& b, const Tensor& res, double alpha, double p, int64_t site, int64_t rs) {
        return PwResFn::apply(x, w, b, res, alpha, p, site, rs);
    });
    m.def("pw_res_ok", [](const Tensor& x, const Tensor& w) { return pw_gelu_fusable(x, w); });
    m.def("qkv", [](const Tensor& x, const Tensor& wq, const OptT& bq, const Tensor& wk, const OptT& bk, const Tensor& wv, const OptT& bv) {
        return QKVFn::apply(x, wq, bq, wk, bk, wv, bv);
    });
    // weight images of a JLC block ahead of its forward: jlc_img_plan -> (kind, floats) for a block of C channels / G groups on a D x H x W grid (kind 0: the block builds
    // nothing ahead); jlc_prep_into builds them into `img` on `stream` and leaves them for the block's next forward (jlc_fwd_f); jlc_prefetch_clear drops what was not taken
    m.def("jlc_img_plan", [](int64_t C, int64_t G, int64_t D, int64_t H, int64_t W) {
        const int kind = jlc_img_kind((int)C, (int)G, (int)D, (int)H, (int)W);
        const long n = kind == 1 ? (long)vx_jlc_tz_img_floats_ns((int)C, (int)G, vx_jlc_tz_pieces()) : kind == 2 ? (long)vx_jlc_cl_img_floats((int)C, (int)G) : 0;
        return std::make_pair((int64_t)kind, (int64_t)n);
    });
    m.def("weights_epoch_bump", []() { return ++WEIGHTS_EPOCH; });
    m.def("weights_epoch", []() { return WEIGHTS_EPOCH; });
    m.def("jlc_prep_into", [](const Tensor& w1, const Tensor& w3, const Tensor& w5, Tensor img, int64_t C, int64_t G, int64_t D, int64_t H, int64_t W, int64_t stream, bool keep) {
        const int kind = jlc_img_kind((int)C, (int)G, (int)D, (int)H, (int)W);
        if (kind == 0) return false;
        void* s_ = sp(stream);
        JlcPreImg q;
        q.img = img; q.kind = kind; q.C = (int)C; q.G = (int)G;
        if (kind == 1) {
            q.pieces = vx_jlc_tz_pieces();
            TORCH_CHECK(img.numel() >= (long)vx_jlc_tz_img_floats_ns((int)C, (int)G, q.pieces), "jlc_prep_into: image buffer too small");
            VX(vx_jlc_tz_prep_ns, fp(w1), fp(w3), fp(w5), mp(img), (int)C, (int)G, q.pieces, s_);
        } else {
            TORCH_CHECK(img.numel() >= (long)vx_jlc_cl_img_floats((int)C, (int)G), "jlc_prep_into: image buffer too small");
            VX(vx_jlc_cl_prep, fp(w1), fp(w3), fp(w5), mp(img), (int)C, (int)G, s_);
        }
        q.keep = keep; q.epoch = WEIGHTS_EPOCH; q.ver = jlc_wver(w1, w3, w5);
        JLC_PRE[w1.data_ptr()] = q;
        return true;
    }, py::arg("w1"), py::arg("w3"), py::arg("w5"), py::arg("img"), py::arg("C"), py::arg("G"), py::arg("D"), py::arg("H"), py::arg("W"), py::arg("stream"), py::arg("keep") = false);
    m.def("jlc_prefetch_clear", []() { const int64_t n = (int64_t)(JLC_PRE.size() + EXPAND_PRE.size()); JLC_PRE.clear(); EXPAND_PRE.clear(); return n; });
    // the same for a patch-expand layer (Conv3d k3 p1, 16 -> 64 Cc channels + PixelShuffle(4)) in the fp16-piece mode: -> floats per workspace (0: nothing is built ahead)
    m.def("expand_img_floats", [](int64_t Cout) {
        return (int64_t)((F.use_expand_mfma && !F.bf16_expand && F.expand_split == 22 && F.use_s1 && Cout % 64 == 0) ? vx_expand_split_ws_floats((int)(Cout / 64), 22) : 0);
    });
    m.def("expand_prep_into", [](const Tensor& w, Tensor wt_fwd, Tensor wt_bwd, int64_t stream, bool keep) {
        const int Cc = (int)(w.size(0) / 64);
        if (!(F.use_expand_mfma && !F.bf16_expand && F.expand_split == 22 && F.use_s1) || w.size(0) % 64 != 0 || w.size(1) != 16 || w.size(2) != 3) return false;
        const long n = vx_expand_split_ws_floats(Cc, 22);
        TORCH_CHECK(wt_fwd.numel() >= n && wt_bwd.numel() >= n, "expand_prep_into: workspace too small");
        VX(vx_expand_prep_split22, fp(w), mp(wt_fwd), mp(wt_bwd), Cc, sp(stream));
        ExpandPreImg q;
        q.wt_fwd = wt_fwd; q.wt_bwd = wt_bwd; q.Cc = Cc;
        q.keep = keep; q.epoch = WEIGHTS_EPOCH; q.ver = (int64_t)w._version();
        EXPAND_PRE[w.data_ptr()] = q;
        return true;
    }, py::arg("w"), py::arg("wt_fwd"), py::arg("wt_bwd"), py::arg("stream"), py::arg("keep") = false);
    m.def("jlc", [](const Tensor& x, const Tensor& w0, const OptT& w1, const OptT& w2, const Tensor& b0, const OptT& b1, const OptT& b2, int64_t G, const Tensor& l1w,
                    const Tensor& l1b, const Tensor& l2w, const Tensor& l2b, double p, int64_t site, int64_t rs) {
        return JLCFn::apply(x, w0, w1, w2, b0, b1, b2, G, l1w, l1b, l2w, l2b, p, site, rs);
    });
    m.def("ffn", [](const Tensor& y, const Tensor& gamma, const Tensor& beta, const Tensor& w1, const Tensor& b1, const Tensor& w2, const Tensor& b2, double p, int64_t site1,
                    int64_t site2, int64_t rs) { return FFNFn::apply(y, gamma, beta, w1, b1, w2, b2, p, site1, site2, rs); });
    m.def("layernorm", [](const Tensor& x, const Tensor& g, const Tensor& bt) { return LayerNormFn::apply(x, g, bt); });
    m.def("upconv_k2s2", [](const Tensor& x, const Tensor& w, const OptT& b, bool feeds_in) { return UpconvFn::apply(x, w, b, feeds_in); }, py::arg("x"), py::arg("w"), py::arg("b"),
          py::arg("feeds_in") = false);
    m.def("space_to_depth2", [](const Tensor& x) { return S2DFn::apply(x); });
    m.def("upsample_trilinear", [](const Tensor& x, int64_t D, int64_t H, int64_t W) { return UpsampleFn::apply(x, D, H, W); });
    m.def("gram", [](const Tensor& x) { return GramFn::apply(x); });
    m.def("pwa_core", [](const Tensor& table, int64_t plan_ptr, int64_t cq, int64_t cv, double p_attn, int64_t site, int64_t rs, std::vector<Tensor> qkv) {
        return PwaCoreFn::apply(table, plan_ptr, cq, cv, p_attn, site, rs, at::TensorList(qkv));
    });
    m.def("gelu", [](const Tensor& a, double p, int64_t site, int64_t rs) { return GeluFn::apply(a, p, site, rs); });
    m.def("axpy", [](const OptT& x, const Tensor& z, double alpha, double p, int64_t site, int64_t rs) { return AxpyFn::apply(x, z, alpha, p, site, rs); });

    // weight-gradient side stream: enable around the backward pass, join (make `stream` wait for it) before anything reads the parameter gradients
    m.def("set_wgrad_stream", [](bool on) { WG.enabled = on; });
    // deferral on the submitting streams (taped encoder backward): the weight-gradient kernels of a stream run after everything else queued on it, so
    // the input gradients that OTHER streams wait for leave it earlier; no extra stream competes for CUs.  The closures keep their operands alive.
    m.def("set_wgrad_defer", [](bool on) { WG.enabled = on; WG.same = on; });
    m.def("set_wgrad_hold", []() { WG.enabled = false; WG.same = false; });      // stop deferring but KEEP what is queued (a later wgrad_join under set_wgrad_defer launches it)
    // a weight-gradient launch written in python (functional._submit_wgrad: the grouped launch of the fused PWA chains): same queue, same rules.  The
    // callable receives the stream to launch on; it is released under the GIL wherever the queue drops it.
    m.def("wgrad_submit_py", [](int64_t stream, int64_t device, py::function f) {
        std::shared_ptr<py::function> sf(new py::function(std::move(f)), [](py::function* p) { py::gil_scoped_acquire g; delete p; });
        wgrad_submit(sp(stream), (int)device, [sf](void* s) { py::gil_scoped_acquire g; (*sf)((int64_t)(uintptr_t)s); });
    });
    // spread > 1 (deferral on the submitting streams only): the queued launches are independent sinks, so instead of running one after the other at the
    // end of the stream they were queued from they are dealt round-robin onto `spread` streams (the submitting stream and spread - 1 forked ones, which
    // first wait for it); the joining stream waits for all of them.  In a captured stage these become parallel branches, i.e. different tape lanes.
    m.def("wgrad_deferring", []() { return WG.enabled; });
    m.def("wgrad_join", [](int64_t stream, int64_t device, bool final, int64_t spread) {
        if (WG.same) {
            hipStream_t js = (hipStream_t)sp(stream);
            if (WG.ev[0] == nullptr) for (auto& e : WG.ev) TORCH_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess, "hipEventCreate failed");
            static std::vector<c10::hip::HIPStreamMasqueradingAsCUDA> sides;
            const int nsp = (int)std::max<int64_t>(1, std::min<int64_t>(spread, 4));
            while ((int)sides.size() < nsp - 1) sides.push_back(c10::hip::getStreamFromPoolMasqueradingAsCUDA(false, (c10::DeviceIndex)device));
            void* seen[8]; int ns = 0;
            std::vector<std::pair<void*, int>> forked;                    // (submitting stream, side index) pairs already ordered
            bool side_used[4] = {false, false, false, false};
            for (size_t i = 0; i < WG.pending.size(); ++i) {
                auto& p = WG.pending[i];
                const int k = nsp > 1 ? (int)(i % (size_t)nsp) : 0;
                const c10::DeviceIndex dev = (c10::DeviceIndex)(i < WG.pending_dev.size() ? WG.pending_dev[i] : device);
                if (k == 0) {
                    // the closure allocates its temporaries (partial-sum slices, a channels-last copy) when it is launched.  They must come from
                    // the pool of the stream the kernels run on: a cached block of the CALLER's stream may still be in use by kernels queued on
                    // it, which the launch stream does not wait for (seen as a corrupted level-1 attention backward, brats128 B=4, eager stages)
                    auto ls = p.first ? c10::hip::getStreamFromExternalMasqueradingAsCUDA((hipStream_t)p.first, dev) : c10::hip::getDefaultHIPStreamMasqueradingAsCUDA(dev);
                    c10::hip::HIPStreamGuardMasqueradingAsCUDA guard(ls);
                    p.second(p.first);
                    bool dup = p.first == (void*)js;
                    for (int j = 0; j < ns; ++j) dup = dup || seen[j] == p.first;
                    if (!dup && ns < 8) seen[ns++] = p.first;
                } else {
                    auto& side = sides[k - 1];
                    bool ordered = false;
                    for (auto& f : forked) ordered = ordered || (f.first == p.first && f.second == k);
                    if (!ordered) { wg_order((hipStream_t)p.first, side.stream()); forked.emplace_back(p.first, k); }
                    c10::hip::HIPStreamGuardMasqueradingAsCUDA guard(side);
                    p.second((void*)side.stream());
                    side_used[k] = true;
                }
            }
            for (int j = 0; j < ns; ++j) wg_order((hipStream_t)seen[j], js);      // the joining stream waits for the streams that got late work
            for (int k = 1; k < nsp; ++k) if (side_used[k]) wg_order(sides[k - 1].stream(), js);
            WG.pending.clear(); WG.pending_dev.clear();
            WG.done.clear();
            return;
        }
        wg_flush((int)device);
        if (WG.stream.has_value() && !WG.done.empty()) wg_order(WG.stream->stream(), (hipStream_t)sp(stream));
        if (final) WG.done.clear();          // after the wait above: the memory may be reused by the joining stream
    }, py::arg("stream"), py::arg("device"), py::arg("final"), py::arg("spread") = 0);

    // kernel pass of bench.py: every C-ABI call made from this module between profile_begin() and profile_end() is timed with HIP events on its
    // launch stream; profile_end() -> [(entry, (small integer arguments...), ms)]
    m.def("profile_begin", []() { PROF.recs.clear(); PROF.on = true; });
    m.def("profile_end", []() {
        PROF.on = false;
        TORCH_CHECK(hipDeviceSynchronize() == hipSuccess, "hipDeviceSynchronize failed");
        py::list out;
        for (auto& r : PROF.recs) {
            float ms = 0.0f;
            hipEventElapsedTime(&ms, r.e0, r.e1);
            hipEventDestroy(r.e0); hipEventDestroy(r.e1);
            py::tuple key(r.key.size());
            for (size_t i = 0; i < r.key.size(); ++i) key[i] = py::int_(r.key[i]);
            out.append(py::make_tuple(std::string(r.name), key, ms));
        }
        PROF.recs.clear();
        return out;
    });
    m.def("set_conv_mfma", [](bool on) { F.use_conv_mfma = on; });
    m.def("set_bf16_expand", [](bool on) { F.bf16_expand = on; });      // bf16 opt-in mode: bf16 MFMA operands in the patch-expand forward / input gradient
    m.def("get_bf16_expand", []() { return F.bf16_expand; });
    m.def("set_expand_split", [](int64_t ns) { F.expand_split = (ns == 2 || ns == 3 || ns == 22) ? (int)ns : 0; });      // fp32 mode: split products in the patch-expand layers (2/3 bf16 pieces, 22 = two scaled fp16 pieces; 0 = fp32 MFMA)
    m.def("get_expand_split", []() { return F.expand_split; });
    m.def("set_tile_min_c", [](int64_t c) { F.tile_min_c = (int)c; });
    m.def("set_act_bf16", [](bool on) { F.act_bf16 = on; }, "bf16 STORAGE mode: block-internal tensors of the JLC blocks / full-resolution heads and their gradients as bf16 arrays (with set_bf16_expand and vx_jlc_tz_set_pieces(1): functional.set_precision('bf16'))");
    m.def("get_act_bf16", []() { return F.act_bf16; });
    m.def("set_upconv_wgrad_mfma", [](bool on) { F.upconv_wgrad_mfma = on; });
    m.def("set_jlc_tz", [](bool on) { F.jlc_tz = on; });          // A/B (tests): JLC grouped convs on the matrix pipe (default) or the fp32 VALU kernels
    m.def("get_jlc_tz", []() { return F.jlc_tz; });
    m.def("set_jlc_wg_tz", [](bool on) { F.jlc_wg_tz = on; });    // A/B (tests): JLC weight gradients on the matrix pipe (default) or the VALU kernels
    m.def("set_jlc_tile", [](bool on) { F.jlc_tile = on; });      // A/B (tests): JLC block of the coarse levels fused (default) or per operator
    m.def("set_expand_wgrad_split", [](bool on) { F.expand_wgrad_split = on; });      // A/B (tests, probes): weight gradient of the patch-expand layers on the split kernels
    m.def("set_fuse_blocks", [](bool on) { F.fuse_blocks = on; });     // A/B: JLC block / FFN tail on the fused block kernels (jlc.hip, mlp.hip) vs the per-operator kernels
    m.def("set_fuse_gelu", [](bool on) { F.fuse_gelu = on; });
    m.def("set_fuse_bwd_add", [](bool on) { F.fuse_bwd_add = on; });   // A/B: residual-gradient sums in the stores of the InstanceNorm / LayerNorm backward kernels
    m.def("set_fuse_res", [](bool on) { F.fuse_res = on; });           // A/B: residual + dropout in the epilogue of the second 1x1 conv of the JLC / FFN stage
    m.def("set_in_split2", [](bool on) { F.in_split2 = on; });         // A/B: long-row InstanceNorm in 2 launches per direction (0 = separate stats / finalise / apply launches)
    m.def("set_skip_in_bias", [](bool on) { F.skip_in_bias = on; });   // A/B: 0 = compute the (mathematically zero) bias gradients of convs that feed an InstanceNorm
    m.def("set_down_mfma", [](bool on) { F.use_down_mfma = on; });  // A/B: MFMA weight gradient of the k7 s4 stem conv
    m.def("set_fuse_pw_bwd", [](bool on) { F.fuse_pw_bwd = on; });  // A/B: input + weight gradient of small 1x1 convs in one launch      // A/B: GELU (+ dropout) in the 1x1 conv epilogues of the JLC / FFN composites

    m.def("conv_fwd", [](const Tensor& x, const OptT& x2, const Tensor& w, const OptT& b, int K, int S, int P, int G, int ps, int64_t stream) {
        auto st = std::make_shared<ConvState>();
        Tensor y = conv_fwd_impl(*st, x, x2.value_or(Tensor()), w, b.value_or(Tensor()), K, S, P, G, ps, x.requires_grad(), sp(stream));
        return py::make_tuple(y, st);
    });
    m.def("conv_bwd", [](std::shared_ptr<ConvState> st, const Tensor& dy, bool need_x, int64_t stream) {
        Tensor dx, dx2;
        conv_bwd_impl(*st, dy, need_x, dx, dx2, sp(stream));
        return py::make_tuple(dx.defined() ? py::cast(dx) : py::none(), dx2.defined() ? py::cast(dx2) : py::none());
    });

    m.def("in_fwd", [](const OptT& res, bool act, const std::vector<Tensor>& ys, int64_t stream) {
        auto st = std::make_shared<INState>();
        Tensor out = in_fwd_impl(*st, res.value_or(Tensor()), act, ys, sp(stream));
        return py::make_tuple(out, st);
    });
    m.def("in_bwd", [](std::shared_ptr<INState> st, const Tensor& dout, const std::vector<bool>& need, int64_t stream) {
        auto g = in_bwd_impl(*st, dout, need, sp(stream));
        py::list out;
        for (auto& t : g) out.append(t.defined() ? py::cast(t) : py::none());
        return out;
    });

    m.def("ln_fwd", [](const Tensor& x, const Tensor& g, const Tensor& bt, int64_t stream) {
        auto st = std::make_shared<LNState>();
        Tensor out = ln_fwd_impl(*st, x, g, bt, sp(stream));
        return py::make_tuple(out, st);
    });
    m.def("ln_bwd", [](std::shared_ptr<LNState> st, const Tensor& dout, int64_t stream) { return ln_bwd_impl(*st, dout, sp(stream)); });

    m.def("gelu_fwd", [](const Tensor& a, double p, int64_t site, int64_t rs, int64_t stream) {
        auto st = std::make_shared<GeluState>();
        Tensor h = gelu_fwd_impl(*st, a, p, site, sp(rs), sp(stream));
        return py::make_tuple(h, st);
    });
    m.def("gelu_bwd", [](std::shared_ptr<GeluState> st, const Tensor& dh, int64_t stream) { return gelu_bwd_impl(*st, dh, sp(stream)); });

    m.def("axpy_fwd", [](const OptT& x, const Tensor& z, double alpha, double p, int64_t site, int64_t rs, int64_t stream) {
        auto st = std::make_shared<AxpyState>();
        Tensor out = axpy_fwd_impl(*st, x.value_or(Tensor()), z, alpha, p, site, sp(rs), sp(stream));
        return py::make_tuple(out, st);
    });
    m.def("axpy_bwd", [](std::shared_ptr<AxpyState> st, const Tensor& dout, bool need_x, int64_t stream) {
        Tensor dx, dz;
        axpy_bwd_impl(*st, dout, need_x, dx, dz, sp(stream));
        return py::make_tuple(dx.defined() ? py::cast(dx) : py::none(), dz);
    });

    // JLC block: ws/bs = the spatial convs' weights / biases (kernel sizes from the weight shapes), groups G; l1/l2 = the channel MLP
    m.def("jlc_fwd", [](const Tensor& x, const std::vector<Tensor>& ws, const std::vector<Tensor>& bs, int G, const Tensor& l1w, const Tensor& l1b, const Tensor& l2w,
                        const Tensor& l2b, double p, int64_t site, int64_t rs, int64_t stream) {
        auto r = jlc_fwd_f(x, ws, bs, G, l1w, l1b, l2w, l2b, p, site, rs, stream);
        return py::make_tuple(r.first, r.second);
    });
    m.def("jlc_bwd", [](std::shared_ptr<JLCState> st, const Tensor& dout, bool need_x, int64_t stream) -> py::object {
        Tensor d = jlc_bwd_f(st, dout, need_x, stream);
        return d.defined() ? py::cast(d) : py::none();
    });

    // FFN tail: out = y + Drop(linear2(Drop(GELU(linear1(LN(y))))))
    m.def("ffn_fwd", [](const Tensor& y, const Tensor& gamma, const Tensor& beta, const Tensor& w1, const Tensor& b1, const Tensor& w2, const Tensor& b2, double p,
                        int64_t site1, int64_t site2, int64_t rs, int64_t stream) {
        auto r = ffn_fwd_f(y, gamma, beta, w1, b1, w2, b2, p, site1, site2, rs, stream);
        return py::make_tuple(r.first, r.second);
    });
    m.def("ffn_bwd", [](std::shared_ptr<FFNState> st, const Tensor& dout, int64_t stream) { return ffn_bwd_f(st, dout, stream); });
}
