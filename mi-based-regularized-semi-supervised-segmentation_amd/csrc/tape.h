// Launch tape: the library records its own entry-point calls and issues them again from ONE C call (csrc/tape.hip).
//
// The reference's hot loop (semi_seg/epocher.py:143-187) is Python; so is this repo's.  One udaiic iteration is ~300 kernel launches
// on three HIP streams, issued from two Python threads (the caller's and the autograd engine's) at 15-25 us of interpreter time each:
// on a box with slow host cores the step is host-bound.  Every launch goes through an `extern "C"` entry point with plain-old-data
// arguments, so an iteration with fixed shapes is a fixed list of (entry point, arguments, stream): MISEG_TAPE at the head of each
// launching entry point appends the call to the tape that is being recorded (the outermost call only -- entry points that forward
// to others record once), and miseg_tape_replay calls the list again, in the recorded order, hence with the recorded per-stream
// order and cross-stream waits: the GPU sees what eager submission gave it, the host spends ~3 us per launch instead of ~20.
// Pointer arguments that change between iterations (the loader's batch tensors, the pinned staging slots, the read-back event) are
// BOUND: miseg_tape_bind(base, span) finds every recorded pointer inside [base, base + span) and miseg_tape_replay re-bases them.
#pragma once
#include <stdint.h>

#include <functional>
#include <memory>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

namespace miseg_core {

struct TapePtrRef {
    const void** loc;      // where the recorded pointer lives (inside the op's argument tuple)
};

struct TapeOpBase {
    const char* name = "";
    void* stream = nullptr;
    virtual ~TapeOpBase() {}
    virtual int run() = 0;
    virtual void pointers(std::vector<TapePtrRef>& out) = 0;
};

template <class... A> struct TapeOp final : TapeOpBase {
    int (*fn)(A...);
    std::tuple<A...> args;
    TapeOp(int (*f)(A...), A... a) : fn(f), args(a...) {}
    int run() override { return std::apply(fn, args); }
    void pointers(std::vector<TapePtrRef>& out) override {
        std::apply([&](auto&... e) { (collect(out, e), ...); }, args);
    }
    template <class T> static void collect(std::vector<TapePtrRef>& out, T& e) {
        if constexpr (std::is_pointer_v<T>) out.push_back({(const void**)(void*)&e});      // any object pointer type: same size and representation
    }
};

struct TapeFnOp final : TapeOpBase {
    std::function<int()> fn;
    std::vector<const void**> ptrs;      // the pointers the closure reads through (bindable)
    int run() override { return fn(); }
    void pointers(std::vector<TapePtrRef>& out) override {
        for (auto p : ptrs) out.push_back({p});
    }
};

extern thread_local int tape_depth;       // > 0 inside an entry point (or a replay) of this thread
bool tape_recording();                    // a tape is being recorded (process-wide: the backward pass runs on another thread)
void tape_push(TapeOpBase* op);           // takes ownership

struct TapeScope {
    TapeScope() { ++tape_depth; }
    ~TapeScope() { --tape_depth; }
};

template <class... A, class... B> inline void tape_record(const char* name, int (*fn)(A...), void* stream, B... rest) {
    auto* op = new TapeOp<A...>(fn, stream, rest...);
    op->name = name;
    op->stream = stream;
    tape_push(op);
}

}  // namespace miseg_core

// First statement of every launching entry point (first argument = the stream).  The f16 twins (MISEG_F16_BUILD) are only ever reached
// through their primaries, which have recorded the call already.
#ifndef MISEG_F16_BUILD
#define MISEG_TAPE(fn, ...)                                                                        \
    ::miseg_core::TapeScope miseg_tape_scope_;                                                     \
    do {                                                                                           \
        if (::miseg_core::tape_depth == 1 && ::miseg_core::tape_recording()) ::miseg_core::tape_record(#fn, fn, __VA_ARGS__); \
    } while (0)
#else
#define MISEG_TAPE(fn, ...) do { } while (0)
#endif
