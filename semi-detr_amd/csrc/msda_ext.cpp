// Compiled Python module `MultiScaleDeformableAttention` -- the drop-in for the reference's pybind module
// (detr_od/models/utils/ops/src/vision.cpp:13-16, signatures of src/ms_deform_attn.h:20-61, host code of
// src/cuda/ms_deform_attn_cuda.cu:20-153): same two entry points, same argument order, same precondition errors
// (AT_ASSERTM -> RuntimeError), outputs freshly allocated on the inputs' device, launches on the CURRENT stream, no
// synchronisation.  It holds no kernels: it validates at::Tensors and calls the C ABI of libsemidetr_hip.so
// (include/semidetr_hip.h), which is where the hand-written gfx950 code lives.  Host-only C++ (g++), built by
// csrc/Makefile; a ctypes call of the same ABI cost ~16 us per forward against a 4.7 us kernel (BENCH_r01), this
// path costs the launch plus ~2 us.
//
// Besides the reference's two functions it exports the fused MSDeformAttn prologue/epilogue pair (SURVEY.md
// section 8(f) row 1) and `pyramid_check`, the cached test behind SEMIDETR_MSDA_QUERIES_ARE_PIXELS.
#include <torch/extension.h>

// ROCm builds of torch present their devices as DeviceType::CUDA; these are the guard / stream types for that view
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <mutex>
#include <vector>

#include "semidetr_hip.h"

namespace {

void check_rc(int rc, const char *what)
{
    TORCH_CHECK(rc == 0, what, " failed (code ", rc, "): ", semidetr_last_error());
}

const char *scalar_name(at::ScalarType t) { return c10::toString(t); }

// ms_deform_attn.h:27-38 (CPU tensors only raise) and ms_deform_attn_cuda.cu:28-38 / :93-105
void check_tensor(const at::Tensor &t, const at::Tensor &value, const char *name)
{
    TORCH_CHECK(t.is_contiguous(), name, " tensor has to be contiguous");
    TORCH_CHECK(t.is_cuda(), name, " must be a CUDA tensor");
    TORCH_CHECK(t.device() == value.device(), name, " must be on the same device as value");
}

struct Dims {
    int N, S, M, D, L, Lq, P;
};

Dims op_dims(const at::Tensor &value, const at::Tensor &shapes, const at::Tensor &starts, const at::Tensor &loc,
             const at::Tensor &attn, int64_t im2col_step, const char *what)
{
    TORCH_CHECK(value.is_cuda(), "Not implemented on the CPU");
    check_tensor(value, value, "value");
    check_tensor(shapes, value, "spatial_shapes");
    check_tensor(starts, value, "level_start_index");
    check_tensor(loc, value, "sampling_loc");
    check_tensor(attn, value, "attn_weight");
    TORCH_CHECK(value.dim() == 4 && loc.dim() == 6 && attn.dim() == 5, what,
                ": expected value (N,S,M,D), sampling_loc (N,Lq,M,L,P,2), attn_weight (N,Lq,M,L,P)");
    Dims d;
    d.N = (int)value.size(0); d.S = (int)value.size(1); d.M = (int)value.size(2); d.D = (int)value.size(3);
    d.L = (int)shapes.size(0); d.Lq = (int)loc.size(1); d.P = (int)loc.size(4);
    const int64_t step = std::min<int64_t>(d.N, im2col_step);           // ms_deform_attn_cuda.cu:50-52
    TORCH_CHECK(step > 0 && d.N % step == 0, "batch(", d.N, ") must divide im2col_step(", step, ")");
    // the reference reads .data<int64_t>() of both index tensors -> anything but int64 raises
    TORCH_CHECK(shapes.scalar_type() == at::kLong, "expected scalar type Long for spatial_shapes");
    TORCH_CHECK(starts.scalar_type() == at::kLong, "expected scalar type Long for level_start_index");
    TORCH_CHECK(value.scalar_type() == at::kFloat || value.scalar_type() == at::kDouble, "\"", what,
                "\" not implemented for '", scalar_name(value.scalar_type()), "'");
    TORCH_CHECK(loc.scalar_type() == value.scalar_type() && attn.scalar_type() == value.scalar_type(), what,
                ": value, sampling_loc and attn_weight must share one dtype");
    TORCH_CHECK(loc.size(0) == d.N && loc.size(2) == d.M && loc.size(3) == d.L && loc.size(5) == 2 &&
                    attn.size(0) == d.N && attn.size(1) == d.Lq && attn.size(2) == d.M && attn.size(3) == d.L &&
                    attn.size(4) == d.P && starts.numel() == d.L && shapes.dim() == 2 && shapes.size(1) == 2,
                what, ": inconsistent tensor shapes");
    return d;
}

// ---- is the level table an exact tiling of [0, S)?  The answer needs the table on the host (one blocking copy), so
// it is cached per (spatial_shapes, level_start_index) tensor pair and version: a training step passes the same two
// tensors to all twelve layers, so this is at most one synchronisation per step where the reference's module has one
// per layer call (`assert (...).sum() == Len_in`, modules/ms_deform_attn.py:90).
struct PyramidEntry {
    c10::weak_intrusive_ptr<c10::TensorImpl> shapes, starts;
    uint32_t v_shapes, v_starts;
    const void *p_shapes, *p_starts;
    int64_t S;
    int result;
};
std::mutex g_pyr_mutex;
std::vector<PyramidEntry> g_pyr_cache;

// bit 0: sum(H*W) == S (the reference's assert); bit 1: level_start tiles [0, S) contiguously
int pyramid_check(const at::Tensor &shapes, const at::Tensor &starts, int64_t S)
{
    auto *is = shapes.unsafeGetTensorImpl();
    auto *it = starts.unsafeGetTensorImpl();
    // Inference tensors have no version counter (torch.inference_mode(): "Inference tensors do not track version
    // counter"): the cache cannot tell whether they changed, so they are checked on every call.
    const bool cacheable = !shapes.is_inference() && !starts.is_inference();
    if (cacheable) {
        std::lock_guard<std::mutex> lock(g_pyr_mutex);
        for (auto &e : g_pyr_cache) {
            if (e.S != S || e.shapes._unsafe_get_target() != is || e.starts._unsafe_get_target() != it) continue;
            auto a = e.shapes.lock();      // expired (address reused by a new tensor) -> nullptr
            auto b = e.starts.lock();
            // also keyed on the storage address: `t.data = other` re-points a tensor without touching its version.
            // (In-place writes through `t.data` itself bump no counter anyone can see -- as for torch's own autograd
            // checks, that is outside what a version-based cache can notice.)
            if (a && b && e.v_shapes == is->version_counter().current_version() &&
                e.v_starts == it->version_counter().current_version() && e.p_shapes == shapes.data_ptr() &&
                e.p_starts == starts.data_ptr())
                return e.result;
        }
    }
    TORCH_CHECK(shapes.scalar_type() == at::kLong && starts.scalar_type() == at::kLong && shapes.dim() == 2 &&
                    shapes.size(1) == 2 && starts.numel() == shapes.size(0),
                "pyramid_check: expected spatial_shapes (L,2) and level_start_index (L,) of dtype int64");
    if (shapes.is_cuda()) {      // a table not seen before costs a device-to-host copy, which a stream capture does not allow:
        // answer "nothing known" (the kernels that assume nothing) and do not cache it
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        hipStream_t cur = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(shapes.device().index()).stream();
        if (hipStreamIsCapturing(cur, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return 0;
    }
    const at::Tensor hs = shapes.to(at::kCPU).contiguous(), hl = starts.to(at::kCPU).contiguous();
    const int64_t *ps = hs.data_ptr<int64_t>(), *pl = hl.data_ptr<int64_t>();
    int64_t sum = 0;
    bool tiles = true;
    for (int64_t l = 0; l < hs.size(0); ++l) {
        // (height / width <= 32766: the region-window forward packs a sample's top-left pixel into 15 + 15 bits, semidetr_hip.h)
        tiles = tiles && pl[l] == sum && ps[2 * l] > 0 && ps[2 * l + 1] > 0 && ps[2 * l] <= 32766 && ps[2 * l + 1] <= 32766;
        sum += ps[2 * l] * ps[2 * l + 1];
    }
    const int result = (sum == S ? 1 : 0) | (tiles && sum == S ? 2 : 0);
    if (!cacheable) return result;
    std::lock_guard<std::mutex> lock(g_pyr_mutex);
    if (g_pyr_cache.size() >= 16) g_pyr_cache.erase(g_pyr_cache.begin());
    g_pyr_cache.push_back({c10::weak_intrusive_ptr<c10::TensorImpl>(shapes.getIntrusivePtr()),
                           c10::weak_intrusive_ptr<c10::TensorImpl>(starts.getIntrusivePtr()),
                           is->version_counter().current_version(), it->version_counter().current_version(),
                           shapes.data_ptr(), starts.data_ptr(), S, result});
    return result;
}

// `flags` of the f32 entry points: the self-attention statement, the call site's slot of the forward-kernel choice, and -- while
// torch.use_deterministic_algorithms(True) is in force -- a forward kernel that does not depend on earlier launches
int self_attention_flags(const at::Tensor &shapes, const at::Tensor &starts, int Lq, int S, int64_t policy_slot = 0)
{
    // (the backward calls may carry, OR-ed into the slot number, what gather_choice(slot) answered when the matching forward ran:
    //  SEMIDETR_MSDA_GATHER_WINDOW / _PATCH, bits 16 / 17 -- the autograd functions do that)
    constexpr int64_t kGather = SEMIDETR_MSDA_GATHER_WINDOW | SEMIDETR_MSDA_GATHER_PATCH;
    TORCH_CHECK(policy_slot >= 0 && (policy_slot & ~kGather) <= 255 && (policy_slot & kGather) != kGather,
                "policy_slot must be 0..255 (optionally | gather_choice(slot)), got ", policy_slot);
    return ((Lq == S && (pyramid_check(shapes, starts, S) & 2)) ? SEMIDETR_MSDA_QUERIES_ARE_PIXELS : 0) |
           SEMIDETR_MSDA_POLICY_SLOT((int)(policy_slot & 0xff)) | (int)(policy_slot & kGather) |
           (at::globalContext().deterministicAlgorithms() ? SEMIDETR_MSDA_FIXED_FORWARD : 0);
}

void *stream_of(const at::Tensor &t) { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }

at::Tensor ms_deform_attn_forward(const at::Tensor &value, const at::Tensor &spatial_shapes,
                                  const at::Tensor &level_start_index, const at::Tensor &sampling_loc,
                                  const at::Tensor &attn_weight, int64_t im2col_step, int64_t policy_slot)
{
    const Dims d = op_dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step,
                           "ms_deform_attn_forward_cuda");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(value.device());
    at::Tensor out = at::empty({d.N, d.Lq, (int64_t)d.M * d.D}, value.options());      // the kernel writes all of it
    if (out.numel() == 0 || value.numel() == 0) return out.zero_();
    const int64_t *sh = spatial_shapes.data_ptr<int64_t>(), *ls = level_start_index.data_ptr<int64_t>();
    int rc;
    if (value.scalar_type() == at::kDouble)
        rc = semidetr_msda_forward_f64(stream_of(value), value.data_ptr<double>(), sh, ls, sampling_loc.data_ptr<double>(),
                                       attn_weight.data_ptr<double>(), d.N, d.S, d.M, d.D, d.L, d.Lq, d.P,
                                       out.data_ptr<double>());
    else
        rc = semidetr_msda_forward_f32(stream_of(value), value.data_ptr<float>(), sh, ls, sampling_loc.data_ptr<float>(),
                                       attn_weight.data_ptr<float>(), d.N, d.S, d.M, d.D, d.L, d.Lq, d.P,
                                       self_attention_flags(spatial_shapes, level_start_index, d.Lq, d.S, policy_slot),
                                       out.data_ptr<float>());
    check_rc(rc, "ms_deform_attn_forward");
    return out;
}

std::vector<at::Tensor> ms_deform_attn_backward(const at::Tensor &value, const at::Tensor &spatial_shapes,
                                                const at::Tensor &level_start_index, const at::Tensor &sampling_loc,
                                                const at::Tensor &attn_weight, const at::Tensor &grad_output,
                                                int64_t im2col_step, int64_t policy_slot)
{
    const Dims d = op_dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step,
                           "ms_deform_attn_backward_cuda");
    check_tensor(grad_output, value, "grad_output");
    TORCH_CHECK(grad_output.scalar_type() == value.scalar_type() &&
                    grad_output.numel() == (int64_t)d.N * d.Lq * d.M * d.D,
                "ms_deform_attn_backward_cuda: grad_output must be (N, Lq, M*D) of value's dtype");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(value.device());
    at::Tensor gv = at::empty_like(value);            // zero-filled inside the call where the kernels need it
    at::Tensor gl = at::empty_like(sampling_loc);     // fully written by the kernels
    at::Tensor ga = at::empty_like(attn_weight);
    if (value.numel() == 0 || gl.numel() == 0) return {gv.zero_(), gl.zero_(), ga.zero_()};
    const int64_t *sh = spatial_shapes.data_ptr<int64_t>(), *ls = level_start_index.data_ptr<int64_t>();
    int rc;
    if (value.scalar_type() == at::kDouble)
        rc = semidetr_msda_backward_f64(stream_of(value), grad_output.data_ptr<double>(), value.data_ptr<double>(), sh, ls,
                                        sampling_loc.data_ptr<double>(), attn_weight.data_ptr<double>(), d.N, d.S, d.M,
                                        d.D, d.L, d.Lq, d.P, gv.data_ptr<double>(), gl.data_ptr<double>(),
                                        ga.data_ptr<double>());
    else
        rc = semidetr_msda_backward_f32(stream_of(value), grad_output.data_ptr<float>(), value.data_ptr<float>(), sh, ls,
                                        sampling_loc.data_ptr<float>(), attn_weight.data_ptr<float>(), d.N, d.S, d.M, d.D,
                                        d.L, d.Lq, d.P, self_attention_flags(spatial_shapes, level_start_index, d.Lq, d.S, policy_slot),
                                        gv.data_ptr<float>(), gl.data_ptr<float>(), ga.data_ptr<float>());
    check_rc(rc, "ms_deform_attn_backward");
    return {gv, gl, ga};
}

// ---- fused prologue / epilogue (not part of the reference's pybind surface) ----------------------------------
bool fused_supported(const at::Tensor &value, const at::Tensor &ref, const at::Tensor &off, const at::Tensor &logits)
{
    return value.is_cuda() && value.scalar_type() == at::kFloat && value.dim() == 4 && value.size(3) == 32 &&
           value.size(2) <= 32 && (ref.size(-1) == 2 || ref.size(-1) == 4) && off.scalar_type() == at::kFloat &&
           logits.scalar_type() == at::kFloat && ref.scalar_type() == at::kFloat;
}

Dims fused_dims(const at::Tensor &value, const at::Tensor &shapes, const at::Tensor &starts, const at::Tensor &ref,
                const at::Tensor &off, const at::Tensor &logits)
{
    TORCH_CHECK(value.is_cuda(), "Not implemented on the CPU");
    check_tensor(value, value, "value");
    check_tensor(shapes, value, "spatial_shapes");
    check_tensor(starts, value, "level_start_index");
    check_tensor(ref, value, "reference_points");
    check_tensor(off, value, "sampling_offsets");
    check_tensor(logits, value, "attention logits");
    TORCH_CHECK(shapes.scalar_type() == at::kLong && starts.scalar_type() == at::kLong,
                "expected scalar type Long for spatial_shapes / level_start_index");
    TORCH_CHECK(value.dim() == 4 && off.dim() == 6 && logits.dim() == 4 && ref.dim() == 4,
                "ms_deform_attn_fused: expected sampling_offsets (N,Lq,M,L,P,2), logits (N,Lq,M,L*P), reference "
                "(N,Lq,L,2|4)");
    TORCH_CHECK(value.scalar_type() == at::kFloat && off.scalar_type() == at::kFloat &&
                    logits.scalar_type() == at::kFloat && ref.scalar_type() == at::kFloat,
                "ms_deform_attn_fused: fp32 tensors only");
    Dims d;
    d.N = (int)value.size(0); d.S = (int)value.size(1); d.M = (int)value.size(2); d.D = (int)value.size(3);
    d.Lq = (int)off.size(1); d.L = (int)off.size(3); d.P = (int)off.size(4);
    TORCH_CHECK(off.size(0) == d.N && off.size(2) == d.M && off.size(5) == 2 && logits.size(0) == d.N &&
                    logits.size(1) == d.Lq && logits.size(2) == d.M && logits.size(3) == (int64_t)d.L * d.P &&
                    ref.size(0) == d.N && ref.size(1) == d.Lq && ref.size(2) == d.L && shapes.size(0) == d.L,
                "ms_deform_attn_fused: inconsistent tensor shapes");
    if (ref.size(-1) != 2 && ref.size(-1) != 4)
        throw py::value_error("Last dim of reference_points must be 2 or 4, but get " + std::to_string(ref.size(-1)) +
                              " instead.");
    return d;
}

// The library reads reference points with 16-byte loads: a contiguous VIEW at an odd 8-byte offset (a batch slice of 2-d points
// with an odd Lq * L) is copied into a fresh, aligned allocation -- small and rare.
at::Tensor aligned16(const at::Tensor &t)
{
    return (reinterpret_cast<uintptr_t>(t.data_ptr()) & 15) == 0 ? t : t.clone();
}

// input_padding_mask (N, S) bool / uint8, True = padding, or None: handed to the kernels as bytes
const unsigned char *mask_ptr(const c10::optional<at::Tensor> &mask, const at::Tensor &value, const Dims &d)
{
    if (!mask.has_value() || !mask->defined()) return nullptr;
    const at::Tensor &m = *mask;
    TORCH_CHECK(m.is_cuda() && m.device() == value.device() && m.is_contiguous() && m.dim() == 2 && m.size(0) == d.N &&
                    m.size(1) == d.S && (m.scalar_type() == at::kBool || m.scalar_type() == at::kByte),
                "ms_deform_attn_fused: padding_mask must be a contiguous (N, S) bool / uint8 tensor on value's device");
    return static_cast<const unsigned char *>(m.data_ptr());
}

// ---- the mask's per-(image, level) extents (include/semidetr_hip.h: semidetr_msda_mask_extents), cached like the pyramid
// check: the reference builds ONE mask per batch and hands it to all twelve layers of a pass (transformer.py:1309,1380), so
// one small launch per batch serves 12 forward + 12 backward calls.  Keyed on the mask / level-table tensors (weak references +
// versions + addresses) and the stream the summary was queued on -- a call on another stream queues its own.
struct ExtentsEntry {
    c10::weak_intrusive_ptr<c10::TensorImpl> mask, shapes;
    uint32_t v_mask, v_shapes;
    const void *p_mask, *p_shapes, *p_starts, *stream;
    at::Tensor ext;
};
std::mutex g_ext_mutex;
std::vector<ExtentsEntry> g_ext_cache;

at::Tensor mask_extents(const at::Tensor &m, const at::Tensor &shapes, const at::Tensor &starts)
{
    TORCH_CHECK(m.is_cuda() && m.is_contiguous() && m.dim() == 2 && (m.scalar_type() == at::kBool || m.scalar_type() == at::kByte),
                "mask_extents: padding_mask must be a contiguous (N, S) bool / uint8 CUDA tensor");
    TORCH_CHECK(shapes.is_cuda() && starts.is_cuda() && shapes.scalar_type() == at::kLong && starts.scalar_type() == at::kLong &&
                    shapes.is_contiguous() && starts.is_contiguous() && shapes.dim() == 2 && shapes.size(1) == 2 &&
                    starts.numel() == shapes.size(0),
                "mask_extents: expected contiguous int64 spatial_shapes (L,2) and level_start_index (L,) on the mask's device");
    // (the kernel indexes mask[n * S + starts[l] + i] for i < H * W -- the level table has to be on the mask's device
    //  and has to describe exactly S pixels; the cached pyramid check answers that without a copy after the first call)
    TORCH_CHECK(shapes.device() == m.device() && starts.device() == m.device(),
                "mask_extents: spatial_shapes / level_start_index must be on the mask's device");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(m.device());
    TORCH_CHECK((pyramid_check(shapes, starts, m.size(1)) & 1) != 0,
                "mask_extents: sum(H * W) of spatial_shapes must equal padding_mask.size(1) (and level_start_index must be readable "
                "outside a stream capture)");
    void *stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(m.device().index()).stream();
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    // (inference tensors carry no version counter; a summary computed inside a capture lives in the graph's memory pool)
    const bool cacheable = !m.is_inference() && !shapes.is_inference() && !capturing;
    auto *im = m.unsafeGetTensorImpl();
    auto *is = shapes.unsafeGetTensorImpl();
    if (cacheable) {
        std::lock_guard<std::mutex> lock(g_ext_mutex);
        for (auto &e : g_ext_cache) {
            if (e.mask._unsafe_get_target() != im || e.shapes._unsafe_get_target() != is || e.stream != stream) continue;
            auto a = e.mask.lock();
            auto b = e.shapes.lock();
            if (a && b && e.v_mask == im->version_counter().current_version() && e.v_shapes == is->version_counter().current_version() &&
                e.p_mask == m.data_ptr() && e.p_shapes == shapes.data_ptr() && e.p_starts == starts.data_ptr())
                return e.ext;
        }
    }
    const int N = (int)m.size(0), S = (int)m.size(1), L = (int)shapes.size(0);
    at::Tensor ext = at::empty({N, L}, m.options().dtype(at::kInt));
    if (ext.numel() == 0) return ext;
    check_rc(semidetr_msda_mask_extents(stream, static_cast<const unsigned char *>(m.data_ptr()), shapes.data_ptr<int64_t>(),
                                        starts.data_ptr<int64_t>(), N, S, L, ext.data_ptr<int>()),
             "mask_extents");
    if (cacheable) {
        std::lock_guard<std::mutex> lock(g_ext_mutex);
        if (g_ext_cache.size() >= 16) g_ext_cache.erase(g_ext_cache.begin());
        g_ext_cache.push_back({c10::weak_intrusive_ptr<c10::TensorImpl>(m.getIntrusivePtr()),
                               c10::weak_intrusive_ptr<c10::TensorImpl>(shapes.getIntrusivePtr()),
                               im->version_counter().current_version(), is->version_counter().current_version(), m.data_ptr(),
                               shapes.data_ptr(), starts.data_ptr(), stream, ext});
    }
    return ext;
}

at::Tensor ms_deform_attn_fused_forward(const at::Tensor &value, const at::Tensor &spatial_shapes,
                                        const at::Tensor &level_start_index, const at::Tensor &reference_points,
                                        const at::Tensor &sampling_offsets, const at::Tensor &attn_logits,
                                        const c10::optional<at::Tensor> &padding_mask, int64_t policy_slot)
{
    const Dims d = fused_dims(value, spatial_shapes, level_start_index, reference_points, sampling_offsets, attn_logits);
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(value.device());
    at::Tensor out = at::empty({d.N, d.Lq, (int64_t)d.M * d.D}, value.options());
    if (out.numel() == 0 || value.numel() == 0) return out.zero_();
    const at::Tensor ref = aligned16(reference_points);
    const unsigned char *mask = mask_ptr(padding_mask, value, d);
    at::Tensor ext;      // (kept alive until the launch is queued; the allocator orders its reuse on this stream)
    if (mask) ext = mask_extents(*padding_mask, spatial_shapes, level_start_index);
    const int rc = semidetr_msda_fused_forward_f32(
        stream_of(value), value.data_ptr<float>(), spatial_shapes.data_ptr<int64_t>(), level_start_index.data_ptr<int64_t>(),
        ref.data_ptr<float>(), (int)reference_points.size(-1), sampling_offsets.data_ptr<float>(),
        attn_logits.data_ptr<float>(), mask, mask ? ext.data_ptr<int>() : nullptr, d.N, d.S, d.M, d.D, d.L, d.Lq, d.P,
        self_attention_flags(spatial_shapes, level_start_index, d.Lq, d.S, policy_slot), out.data_ptr<float>());
    check_rc(rc, "ms_deform_attn_fused_forward");
    return out;
}

std::vector<at::Tensor> ms_deform_attn_fused_backward(const at::Tensor &value, const at::Tensor &spatial_shapes,
                                                      const at::Tensor &level_start_index,
                                                      const at::Tensor &reference_points,
                                                      const at::Tensor &sampling_offsets, const at::Tensor &attn_logits,
                                                      const at::Tensor &grad_output,
                                                      const c10::optional<at::Tensor> &padding_mask, int64_t policy_slot)
{
    const Dims d = fused_dims(value, spatial_shapes, level_start_index, reference_points, sampling_offsets, attn_logits);
    TORCH_CHECK(grad_output.is_contiguous() && grad_output.numel() == (int64_t)d.N * d.Lq * d.M * d.D &&
                    grad_output.scalar_type() == at::kFloat && grad_output.device() == value.device(),
                "ms_deform_attn_fused_backward: grad_output must be a contiguous (N, Lq, M*D) tensor of value's dtype");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(value.device());
    at::Tensor gv = at::empty_like(value), go = at::empty_like(sampling_offsets), gl = at::empty_like(attn_logits);
    if (value.numel() == 0 || go.numel() == 0) return {gv.zero_(), go.zero_(), gl.zero_()};
    const at::Tensor ref = aligned16(reference_points);
    const unsigned char *mask = mask_ptr(padding_mask, value, d);
    at::Tensor ext;
    if (mask) ext = mask_extents(*padding_mask, spatial_shapes, level_start_index);
    const int rc = semidetr_msda_fused_backward_f32(
        stream_of(value), grad_output.data_ptr<float>(), value.data_ptr<float>(), spatial_shapes.data_ptr<int64_t>(),
        level_start_index.data_ptr<int64_t>(), ref.data_ptr<float>(), (int)reference_points.size(-1),
        sampling_offsets.data_ptr<float>(), attn_logits.data_ptr<float>(), mask, mask ? ext.data_ptr<int>() : nullptr, d.N, d.S, d.M,
        d.D, d.L, d.Lq, d.P, self_attention_flags(spatial_shapes, level_start_index, d.Lq, d.S, policy_slot), gv.data_ptr<float>(),
        go.data_ptr<float>(), gl.data_ptr<float>());
    check_rc(rc, "ms_deform_attn_fused_backward");
    return {gv, go, gl};
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "MultiScaleDeformableAttention for MI355X (gfx950): at::Tensor front end of libsemidetr_hip.so";
    // the reference's surface (src/vision.cpp:13-16)
    // (+ an optional trailing `policy_slot`: the call site's slot of the encoder forward-kernel choice, semidetr_hip.h)
    m.def("ms_deform_attn_forward", &ms_deform_attn_forward, "ms_deform_attn_forward", py::arg("value"), py::arg("spatial_shapes"),
          py::arg("level_start_index"), py::arg("sampling_loc"), py::arg("attn_weight"), py::arg("im2col_step"),
          py::arg("policy_slot") = 0);
    m.def("ms_deform_attn_backward", &ms_deform_attn_backward, "ms_deform_attn_backward", py::arg("value"), py::arg("spatial_shapes"),
          py::arg("level_start_index"), py::arg("sampling_loc"), py::arg("attn_weight"), py::arg("grad_output"), py::arg("im2col_step"),
          py::arg("policy_slot") = 0);
    // additions
    m.def("ms_deform_attn_fused_forward", &ms_deform_attn_fused_forward, py::arg("value"), py::arg("spatial_shapes"),
          py::arg("level_start_index"), py::arg("reference_points"), py::arg("sampling_offsets"), py::arg("attn_logits"),
          py::arg("padding_mask") = py::none(), py::arg("policy_slot") = 0);
    m.def("ms_deform_attn_fused_backward", &ms_deform_attn_fused_backward, py::arg("value"), py::arg("spatial_shapes"),
          py::arg("level_start_index"), py::arg("reference_points"), py::arg("sampling_offsets"), py::arg("attn_logits"),
          py::arg("grad_output"), py::arg("padding_mask") = py::none(), py::arg("policy_slot") = 0);
    m.def("fused_supported", &fused_supported);
    m.def("gather_choice", [](int64_t slot) { return semidetr_msda_gather_choice((int)slot); }, py::arg("policy_slot") = 0,
          "What an encoder backward of this slot issued now would pick for grad_sampling_loc / grad_attn_weight (SEMIDETR_MSDA_GATHER_WINDOW "
          "or _PATCH): OR it into the backward call's policy_slot and the backward follows what was known at forward time.");
    m.def("mask_extents", &mask_extents, py::arg("padding_mask"), py::arg("spatial_shapes"), py::arg("level_start_index"),
          "(N, L) int32 words vh | vw << 16: level l of image n is padded exactly on rows >= vh / columns >= vw, or -1.  Cached per "
          "(mask, level table) tensors, version and stream; the fused calls fetch it themselves.");
    m.def("pyramid_check", &pyramid_check,
          "bit 0: sum(H*W) == S; bit 1: level_start_index tiles [0, S) exactly.  Cached per tensor pair / version.");
    m.def("abi_version", []() { return SEMIDETR_ABI_VERSION; });      // the header this front end was compiled against
}
