// The launch tape (tape.h) and the small host-side entry points an all-library iteration needs around its kernels: pinned-memory
// uploads / downloads, events, and byte-moving kernels that replace the torch fills / copies of the step (semi_seg/epocher.py:143-187
// is the loop this serves).  Host code only, except the three trivial kernels at the end.
#include <atomic>
#include <mutex>
#include <string>

#include "common.h"

namespace miseg_core {

thread_local int tape_depth = 0;

struct Binding {
    const char* base;
    int64_t span;
    struct Use { const void** loc; int64_t offset; };
    std::vector<Use> uses;
};

struct TimedOp {
    int64_t op;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pairs;     // one pair per replay since the last read
    size_t used = 0;
};

struct Tape {
    std::vector<std::unique_ptr<TapeOpBase>> ops;
    std::vector<int64_t> marks;            // op index at which segment k+1 starts
    std::vector<Binding> bindings;
    std::vector<TimedOp> timed;
    bool sealed = false;
    ~Tape() {
        for (auto& t : timed)
            for (auto& p : t.pairs) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    }
};

static std::mutex g_mu;
static std::atomic<Tape*> g_rec{nullptr};

bool tape_recording() { return g_rec.load(std::memory_order_acquire) != nullptr; }

void tape_push(TapeOpBase* op) {
    std::lock_guard<std::mutex> lk(g_mu);
    Tape* t = g_rec.load(std::memory_order_relaxed);
    if (t) t->ops.emplace_back(op);
    else delete op;
}

static Tape* as_tape(int64_t h) { return reinterpret_cast<Tape*>(static_cast<intptr_t>(h)); }

}  // namespace miseg_core

using miseg_core::Tape;
using miseg_core::as_tape;

extern "C" int64_t miseg_tape_begin(void) {
    std::lock_guard<std::mutex> lk(miseg_core::g_mu);
    if (miseg_core::g_rec.load()) return 0;
    Tape* t = new Tape();
    miseg_core::g_rec.store(t, std::memory_order_release);
    return static_cast<int64_t>(reinterpret_cast<intptr_t>(t));
}

extern "C" int miseg_tape_end(int64_t tape) {
    std::lock_guard<std::mutex> lk(miseg_core::g_mu);
    Tape* t = as_tape(tape);
    if (!t || miseg_core::g_rec.load() != t) return miseg_core::fail(MISEG_E_INVALID, "tape_end: this tape is not the one being recorded");
    miseg_core::g_rec.store(nullptr, std::memory_order_release);
    t->sealed = true;
    return MISEG_OK;
}

extern "C" int miseg_tape_free(int64_t tape) {
    Tape* t = as_tape(tape);
    if (!t) return MISEG_OK;
    {
        std::lock_guard<std::mutex> lk(miseg_core::g_mu);
        if (miseg_core::g_rec.load() == t) miseg_core::g_rec.store(nullptr, std::memory_order_release);
    }
    delete t;
    return MISEG_OK;
}

extern "C" int64_t miseg_tape_mark(int64_t tape) {
    std::lock_guard<std::mutex> lk(miseg_core::g_mu);
    Tape* t = as_tape(tape);
    if (!t || t->sealed) return -1;
    t->marks.push_back((int64_t)t->ops.size());
    return (int64_t)t->marks.size();          // index of the segment that starts here
}

extern "C" int64_t miseg_tape_len(int64_t tape) {
    std::lock_guard<std::mutex> lk(miseg_core::g_mu);
    Tape* t = as_tape(tape);
    return t ? (int64_t)t->ops.size() : -1;
}

extern "C" int64_t miseg_tape_segments(int64_t tape) {
    Tape* t = as_tape(tape);
    return t ? (int64_t)t->marks.size() + 1 : -1;
}

extern "C" const char* miseg_tape_op_name(int64_t tape, int64_t i) {
    Tape* t = as_tape(tape);
    if (!t || i < 0 || i >= (int64_t)t->ops.size()) return "";
    return t->ops[i]->name;
}

extern "C" int64_t miseg_tape_op_stream(int64_t tape, int64_t i) {
    Tape* t = as_tape(tape);
    if (!t || i < 0 || i >= (int64_t)t->ops.size()) return -1;
    return static_cast<int64_t>(reinterpret_cast<intptr_t>(t->ops[i]->stream));
}

// Every recorded pointer argument inside [base, base + span) becomes re-basable: returns the binding's slot (its position in the
// `values` array of miseg_tape_replay), or < 0.  A binding that matches nothing is still a slot (its value is ignored).
extern "C" int64_t miseg_tape_bind(int64_t tape, const void* base, int64_t span) {
    Tape* t = as_tape(tape);
    if (!t || !t->sealed) return miseg_core::fail(MISEG_E_INVALID, "tape_bind: the tape is still being recorded");
    miseg_core::Binding b;
    b.base = static_cast<const char*>(base);
    b.span = span < 1 ? 1 : span;
    std::vector<miseg_core::TapePtrRef> refs;
    for (auto& op : t->ops) {
        refs.clear();
        op->pointers(refs);
        for (auto& r : refs) {
            const char* p = static_cast<const char*>(*r.loc);
            if (p && p >= b.base && p < b.base + b.span) b.uses.push_back({r.loc, (int64_t)(p - b.base)});
        }
    }
    t->bindings.push_back(std::move(b));
    return (int64_t)t->bindings.size() - 1;
}

extern "C" int64_t miseg_tape_bind_uses(int64_t tape, int64_t slot) {
    Tape* t = as_tape(tape);
    if (!t || slot < 0 || slot >= (int64_t)t->bindings.size()) return -1;
    return (int64_t)t->bindings[slot].uses.size();
}

// HIP-event pair around op `i` at every replay (bench.py's live timing of the dominant kernel, on the stream it is launched on).
extern "C" int miseg_tape_time_op(int64_t tape, int64_t i) {
    Tape* t = as_tape(tape);
    if (!t || i < 0 || i >= (int64_t)t->ops.size()) return miseg_core::fail(MISEG_E_INVALID, "tape_time_op: no such op");
    for (auto& x : t->timed)
        if (x.op == i) return MISEG_OK;
    miseg_core::TimedOp x;
    x.op = i;
    t->timed.push_back(std::move(x));
    return MISEG_OK;
}

// Elapsed times (ms) of the timed op since the last call; synchronises with the recorded events.  Returns how many were written.
extern "C" int64_t miseg_tape_timed_ms(int64_t tape, int64_t i, float* ms, int64_t cap) {
    Tape* t = as_tape(tape);
    if (!t) return -1;
    for (auto& x : t->timed) {
        if (x.op != i) continue;
        int64_t n = 0;
        for (size_t k = 0; k < x.used && n < cap; ++k) {
            float v = 0.f;
            if (hipEventSynchronize(x.pairs[k].second) == hipSuccess && hipEventElapsedTime(&v, x.pairs[k].first, x.pairs[k].second) == hipSuccess) ms[n++] = v;
        }
        x.used = 0;
        return n;
    }
    return 0;
}

extern "C" int miseg_tape_replay(int64_t tape, int64_t segment, const void* const* values, int64_t nvalues) {
    Tape* t = as_tape(tape);
    if (!t || !t->sealed) return miseg_core::fail(MISEG_E_INVALID, "tape_replay: not a sealed tape");
    const int64_t nseg = (int64_t)t->marks.size() + 1;
    if (segment < 0 || segment >= nseg) return miseg_core::fail(MISEG_E_INVALID, "tape_replay: segment %lld of %lld", (long long)segment, (long long)nseg);
    if (nvalues != (int64_t)t->bindings.size()) return miseg_core::fail(MISEG_E_INVALID, "tape_replay: %lld values for %lld bindings", (long long)nvalues, (long long)t->bindings.size());
    if (segment == 0)
        for (int64_t s = 0; s < nvalues; ++s) {
            const char* v = static_cast<const char*>(values[s]);
            for (auto& u : t->bindings[s].uses) *u.loc = v + u.offset;
        }
    const int64_t lo = segment == 0 ? 0 : t->marks[segment - 1];
    const int64_t hi = segment + 1 == nseg ? (int64_t)t->ops.size() : t->marks[segment];
    miseg_core::TapeScope scope;          // nothing below records, whatever another thread is doing
    for (int64_t i = lo; i < hi; ++i) {
        miseg_core::TapeOpBase* op = t->ops[i].get();
        miseg_core::TimedOp* timed = nullptr;
        for (auto& x : t->timed)
            if (x.op == i) timed = &x;
        if (timed) {
            if (timed->used == timed->pairs.size()) {
                hipEvent_t a, b;
                if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return miseg_core::fail(MISEG_E_LAUNCH, "tape_replay: hipEventCreate failed");
                timed->pairs.push_back({a, b});
            }
            hipEventRecord(timed->pairs[timed->used].first, reinterpret_cast<hipStream_t>(op->stream));
        }
        const int rc = op->run();
        if (timed) {
            hipEventRecord(timed->pairs[timed->used].second, reinterpret_cast<hipStream_t>(op->stream));
            ++timed->used;
        }
        if (rc != MISEG_OK) {
            std::string why = miseg_core::last_error_buf();
            return miseg_core::fail(rc, "tape_replay: op %lld (%s) failed: %s", (long long)i, op->name, why.c_str());
        }
    }
    return MISEG_OK;
}

// ---- pinned-memory traffic and events as entry points (so that they are on the tape) -------------------------------------------------
// The iteration's one host -> device copy.  From pinned (device-mapped) host memory it is a KERNEL that reads the host block over the bus:
// hipMemcpyAsync hands a 64 KB host-to-device copy to the DMA engine, and the hand-over between the engines cost the stream ~21 us
// of idle time at every step boundary (profiles/r04d_critical_path: the gap between the download and the step's first kernel).
typedef unsigned int tape_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void upload_kernel(tape_u32x4* __restrict__ dst, const tape_u32x4* __restrict__ src, int n16) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = __builtin_nontemporal_load(src + i);
}

extern "C" int miseg_upload(void* stream, void* dst_dev, const void* src_pinned, int64_t nbytes) {
    MISEG_TAPE(miseg_upload, stream, dst_dev, src_pinned, nbytes);
    if (nbytes <= 0) return MISEG_OK;
    static const bool by_kernel = [] { const char* e = getenv("MISEG_UPLOAD_KERNEL"); return !e || atoi(e) != 0; }();
    hipPointerAttribute_t at;
    if (by_kernel && nbytes % 16 == 0 && (((uintptr_t)dst_dev | (uintptr_t)src_pinned) & 15) == 0 &&
        hipPointerGetAttributes(&at, src_pinned) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer) {
        const int n16 = (int)(nbytes / 16);
        hipLaunchKernelGGL(upload_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           static_cast<tape_u32x4*>(dst_dev), static_cast<const tape_u32x4*>(at.devicePointer), n16);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return miseg_core::fail(MISEG_E_LAUNCH, "upload: %s", hipGetErrorString(e));
        return MISEG_OK;
    }
    (void)hipGetLastError();      // (a pointer the runtime does not know: not an error of this call)
    hipError_t e = hipMemcpyAsync(dst_dev, src_pinned, (size_t)nbytes, hipMemcpyHostToDevice, reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) return miseg_core::fail(MISEG_E_LAUNCH, "upload: %s", hipGetErrorString(e));
    return MISEG_OK;
}

extern "C" int miseg_download(void* stream, void* dst_pinned, const void* src_dev, int64_t nbytes) {
    MISEG_TAPE(miseg_download, stream, dst_pinned, src_dev, nbytes);
    if (nbytes <= 0) return MISEG_OK;
    hipError_t e = hipMemcpyAsync(dst_pinned, src_dev, (size_t)nbytes, hipMemcpyDeviceToHost, reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) return miseg_core::fail(MISEG_E_LAUNCH, "download: %s", hipGetErrorString(e));
    return MISEG_OK;
}

extern "C" int64_t miseg_event_create(void) {
    hipEvent_t ev;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return 0;
    return static_cast<int64_t>(reinterpret_cast<intptr_t>(ev));
}

extern "C" int miseg_event_destroy(int64_t ev) {
    if (ev) hipEventDestroy(reinterpret_cast<hipEvent_t>(static_cast<intptr_t>(ev)));
    return MISEG_OK;
}

// (stream first, like every taped entry point; the event handle is a pointer and can be bound)
extern "C" int miseg_event_record(void* stream, void* ev) {
    MISEG_TAPE(miseg_event_record, stream, ev);
    hipError_t e = hipEventRecord(reinterpret_cast<hipEvent_t>(ev), reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) return miseg_core::fail(MISEG_E_LAUNCH, "event_record: %s", hipGetErrorString(e));
    return MISEG_OK;
}

extern "C" int miseg_event_synchronize(int64_t ev) {
    hipError_t e = hipEventSynchronize(reinterpret_cast<hipEvent_t>(static_cast<intptr_t>(ev)));
    if (e != hipSuccess) return miseg_core::fail(MISEG_E_LAUNCH, "event_synchronize: %s", hipGetErrorString(e));
    return MISEG_OK;
}

extern "C" int miseg_event_query(int64_t ev) {      // 1 = completed, 0 = still pending
    hipError_t e = hipEventQuery(reinterpret_cast<hipEvent_t>(static_cast<intptr_t>(ev)));
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) return 0;
    return miseg_core::fail(MISEG_E_LAUNCH, "event_query: %s", hipGetErrorString(e));
}

// ---- byte movers (kernels, not hipMemcpyDtoD / hipMemsetAsync: those can sit for 100s of us behind a busy device's copy queue) -----
namespace miseg {
__global__ void __launch_bounds__(256) fill_zero_kernel(uint4* __restrict__ dst, int64_t n16, unsigned char* __restrict__ tail, int ntail) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = make_uint4(0, 0, 0, 0);
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}

__global__ void __launch_bounds__(256) copy_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, int64_t n16, unsigned char* __restrict__ dtail,
                                                   const unsigned char* __restrict__ stail, int ntail) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) dtail[threadIdx.x] = stail[threadIdx.x];
}

// out[p] = scale_p[0] * src_p  (or 0 where src_p is null) for up to three consecutive parts of one fp32 buffer: the gradient of the
// logits batch [labeled | unlabeled | flipped unlabeled] assembled from the loss kernels' unscaled gradients and the (device) loss
// coefficients in one pass (ref semi_seg/epocher.py:154-159 splits the batch; autograd would scale, zero-fill and concatenate).
struct RowParts {
    const float* src[3];
    const float* scale[3];
    int64_t end4[3];          // cumulative part ends, in float4 units
};
__global__ void __launch_bounds__(256) assemble_rows_kernel(float4* __restrict__ out, RowParts parts) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t total = parts.end4[2];
    const float s0 = parts.scale[0] ? parts.scale[0][0] : 1.f, s1 = parts.scale[1] ? parts.scale[1][0] : 1.f, s2 = parts.scale[2] ? parts.scale[2][0] : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int p = i < parts.end4[0] ? 0 : (i < parts.end4[1] ? 1 : 2);
        const int64_t begin = p == 0 ? 0 : parts.end4[p - 1];
        const float s = p == 0 ? s0 : (p == 1 ? s1 : s2);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (parts.src[p]) {
            v = reinterpret_cast<const float4*>(parts.src[p])[i - begin];
            v.x *= s; v.y *= s; v.z *= s; v.w *= s;
        }
        out[i] = v;
    }
}
}  // namespace miseg

namespace miseg {
// dst[o][j][:] = src[o][idx[j]][:] (rows of `chunk` floats): the windows of one colour group picked out of a per-window tensor
__global__ void __launch_bounds__(256) gather_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t outer, int64_t n_src, int64_t n_idx,
                                                          const int32_t* __restrict__ idx, int64_t chunk) {
    const int64_t total = outer * n_idx * chunk, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t r = e % chunk, j = (e / chunk) % n_idx, o = e / (chunk * n_idx);
        dst[e] = src[(o * n_src + idx[j]) * chunk + r];
    }
}
}  // namespace miseg

extern "C" int miseg_gather_rows(void* stream, const float* src, float* dst, int64_t outer, int64_t n_src, int64_t n_idx, const int32_t* idx,
                                 int64_t chunk) {
    MISEG_TAPE(miseg_gather_rows, stream, src, dst, outer, n_src, n_idx, idx, chunk);
    MISEG_REQUIRE(src && dst && idx && outer > 0 && n_src > 0 && n_idx > 0 && chunk > 0, "gather_rows: bad args");
    const int64_t total = outer * n_idx * chunk;
    hipLaunchKernelGGL(miseg::gather_rows_kernel, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, 4096))), dim3(256), 0,
                       miseg::as_stream(stream), src, dst, outer, n_src, n_idx, idx, chunk);
    MISEG_LAUNCH_CHECK("gather_rows_kernel");
    return MISEG_OK;
}

static inline int ew_blocks(int64_t n16) { return (int)std::max<int64_t>(1, std::min<int64_t>((n16 + 255) / 256, 2048)); }

extern "C" int miseg_fill_zero(void* stream, void* dst, int64_t nbytes) {
    MISEG_TAPE(miseg_fill_zero, stream, dst, nbytes);
    if (nbytes <= 0) return MISEG_OK;
    MISEG_REQUIRE(dst && (reinterpret_cast<uintptr_t>(dst) & 15) == 0, "fill_zero: destination must be 16-byte aligned");
    const int64_t n16 = nbytes / 16;
    hipLaunchKernelGGL(miseg::fill_zero_kernel, dim3(ew_blocks(n16)), dim3(256), 0, miseg::as_stream(stream), (uint4*)dst, n16,
                       (unsigned char*)dst + n16 * 16, (int)(nbytes - n16 * 16));
    MISEG_LAUNCH_CHECK("fill_zero_kernel");
    return MISEG_OK;
}

extern "C" int miseg_copy(void* stream, void* dst, const void* src, int64_t nbytes) {
    MISEG_TAPE(miseg_copy, stream, dst, src, nbytes);
    if (nbytes <= 0) return MISEG_OK;
    MISEG_REQUIRE(dst && src && ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0, "copy: pointers must be 16-byte aligned");
    const int64_t n16 = nbytes / 16;
    hipLaunchKernelGGL(miseg::copy_kernel, dim3(ew_blocks(n16)), dim3(256), 0, miseg::as_stream(stream), (uint4*)dst, (const uint4*)src, n16,
                       (unsigned char*)dst + n16 * 16, (const unsigned char*)src + n16 * 16, (int)(nbytes - n16 * 16));
    MISEG_LAUNCH_CHECK("copy_kernel");
    return MISEG_OK;
}

extern "C" int miseg_assemble_rows(void* stream, float* out, const float* src0, const float* scale0, int64_t numel0, const float* src1,
                                   const float* scale1, int64_t numel1, const float* src2, const float* scale2, int64_t numel2) {
    MISEG_TAPE(miseg_assemble_rows, stream, out, src0, scale0, numel0, src1, scale1, numel1, src2, scale2, numel2);
    MISEG_REQUIRE(out && numel0 >= 0 && numel1 >= 0 && numel2 >= 0, "assemble_rows: bad arguments");
    MISEG_REQUIRE(numel0 % 4 == 0 && numel1 % 4 == 0 && numel2 % 4 == 0, "assemble_rows: part sizes must be multiples of 4 floats");
    miseg::RowParts parts;
    parts.src[0] = src0; parts.src[1] = src1; parts.src[2] = src2;
    parts.scale[0] = scale0; parts.scale[1] = scale1; parts.scale[2] = scale2;
    parts.end4[0] = numel0 / 4; parts.end4[1] = parts.end4[0] + numel1 / 4; parts.end4[2] = parts.end4[1] + numel2 / 4;
    if (parts.end4[2] == 0) return MISEG_OK;
    hipLaunchKernelGGL(miseg::assemble_rows_kernel, dim3(ew_blocks(parts.end4[2])), dim3(256), 0, miseg::as_stream(stream), (float4*)out, parts);
    MISEG_LAUNCH_CHECK("assemble_rows_kernel");
    return MISEG_OK;
}
