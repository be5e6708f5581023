// Library-level entry points: version, thread-local error message.
#include <stdarg.h>

#include "common.h"

namespace miseg_core {
static thread_local char g_err[512] = "";
char* last_error_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace miseg_core

extern "C" int miseg_version(void) { return 406; }   // round 4 (profiles/r04_pmc.json is keyed by this)
extern "C" const char* miseg_last_error(void) { return miseg::last_error_buf(); }

// ---- cross-stream ordering without a system-scope fence --------------------------------------------------------------------------
// `waiter` waits for everything enqueued on `producer` so far (torch's stream.wait_stream(other)), through a pooled event created with
// hipEventDisableTiming | hipEventDisableSystemFence: the recorded marker then releases at device scope only.  A torch event releases
// to SYSTEM scope -- cache writeback + invalidation across the whole device -- which showed as a ~7 us bubble in front of the next
// kernel of the producing stream at every weight-gradient fork of the backward pass (22 per step; profiles/r03*_critical_path.txt).
// Both streams belong to this process and this device; nothing on the host reads what the producer wrote through this ordering.
namespace miseg_core {
static hipEvent_t next_fork_event() {
    constexpr int kRing = 256, kDev = 16;            // an event may be re-recorded while earlier waits on it are still queued
    struct Ring { hipEvent_t ev[kRing]; int made = 0, next = 0; };
    static thread_local Ring rings[kDev];             // events belong to the device that was current when they were created
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kDev) return nullptr;
    Ring& r = rings[dev];
    constexpr unsigned flags = hipEventDisableTiming | hipEventDisableSystemFence;
    if (r.made < kRing) {
        if (hipEventCreateWithFlags(&r.ev[r.made], flags) != hipSuccess) return nullptr;
        ++r.made;
    }
    hipEvent_t ev = r.ev[r.next % r.made];
    r.next = (r.next + 1) % kRing;
    return ev;
}
}  // namespace miseg_core

extern "C" int miseg_stream_wait_stream(void* waiter, void* producer) {
    MISEG_TAPE(miseg_stream_wait_stream, waiter, producer);
    if (waiter == producer) return MISEG_OK;
    hipEvent_t ev = miseg_core::next_fork_event();
    if (!ev) return miseg_core::fail(MISEG_E_LAUNCH, "stream_wait_stream: hipEventCreateWithFlags failed");
    hipError_t e = hipEventRecord(ev, reinterpret_cast<hipStream_t>(producer));
    if (e == hipSuccess) e = hipStreamWaitEvent(reinterpret_cast<hipStream_t>(waiter), ev, 0);
    if (e != hipSuccess) return miseg_core::fail(MISEG_E_LAUNCH, "stream_wait_stream: %s", hipGetErrorString(e));
    return MISEG_OK;
}
