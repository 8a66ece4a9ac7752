// Library bookkeeping: ABI version + last-error text.
#include <string>

#include "common.h"

namespace dfine {
static thread_local std::string g_last_error;
void set_last_error(hipError_t e) { g_last_error = hipGetErrorString(e); }
}  // namespace dfine

extern "C" {
int dfine_abi_version(void) { return 3; }
const char *dfine_last_error(void) { return dfine::g_last_error.c_str(); }

// Stream `to` waits for everything enqueued on stream `from` so far (fork / join of the side stream that carries the weight
// gradients): one event record + one stream wait, ~2 us of host time against ~15 us for the same through torch.cuda.Stream
// objects, 80 times per step.  Events come from a ring: a stream wait refers to the record that preceded it, so an event can
// be recorded again as soon as its wait has been ENQUEUED.
// A non-blocking stream of the given priority on the current device (hipDeviceGetStreamPriorityRange: -1 high .. 1 low on gfx950;
// PyTorch only hands out 0 and -1).  The weight-gradient side stream is made LOW priority with it: its kernels fill idle CUs but
// must not hold back the main chain's small kernels (a 10 us 1x1 convolution of the backward was measured at up to 199 us behind a
// grid of weight-gradient workgroups of equal priority).  The caller owns the stream (hipStreamDestroy through dfine_stream_destroy).
int dfine_stream_create(int priority, void **out) {
    if (!out) return DFINE_E_BADARG;
    int lo = 0, hi = 0;
    if (hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi); e != hipSuccess) { dfine::set_last_error(e); return DFINE_E_LAUNCH; }
    if (priority > lo) priority = lo;                    // (numerically larger = lower priority)
    if (priority < hi) priority = hi;
    hipStream_t st = nullptr;
    if (hipError_t e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, priority); e != hipSuccess) {
        dfine::set_last_error(e);
        return DFINE_E_LAUNCH;
    }
    *out = st;
    return DFINE_OK;
}

int dfine_stream_destroy(void *stream) {
    if (!stream) return DFINE_OK;
    if (hipError_t e = hipStreamDestroy((hipStream_t)stream); e != hipSuccess) { dfine::set_last_error(e); return DFINE_E_LAUNCH; }
    return DFINE_OK;
}

int dfine_stream_fork(void *from, void *to) {
    constexpr int kRing = 64, kDevices = 16;             // one ring per device: an event belongs to the device it was made on
    static hipEvent_t rings[kDevices][kRing];
    static int nexts[kDevices] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kDevices) return DFINE_E_BADARG;
    hipEvent_t *ring = rings[dev];
    int &next = nexts[dev];
    if (next < 0) {
        for (int i = 0; i < kRing; ++i)
            if (hipError_t e = hipEventCreateWithFlags(&ring[i], hipEventDisableTiming); e != hipSuccess) {
                dfine::set_last_error(e);
                return DFINE_E_LAUNCH;
            }
        next = 0;
    }
    hipEvent_t ev = ring[next];
    next = (next + 1) % kRing;
    hipError_t e = hipEventRecord(ev, (hipStream_t)from);
    if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)to, ev, 0);
    if (e != hipSuccess) {
        dfine::set_last_error(e);
        return DFINE_E_LAUNCH;
    }
    return DFINE_OK;
}

// Host -> device copy of a small table on `stream` from memory the CALLER keeps pinned, alive and unchanged.  Used for the
// pointer tables of launches recorded inside a HIP-graph capture: the copy becomes a memcpy node that re-reads `src` at
// every replay.  (torch's own pinned-memory copies tag the host block with an event for its caching host allocator; an
// event recorded on a capturing stream cannot be queried afterwards - hipErrorCapturedEvent on the next pin_memory().)
int dfine_upload(void *dst, const void *src, int64_t bytes, void *stream) {
    if (bytes == 0) return DFINE_OK;
    if (!dst || !src || bytes < 0) return DFINE_E_BADARG;
    if (hipError_t e = hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, (hipStream_t)stream); e != hipSuccess) {
        dfine::set_last_error(e);
        return DFINE_E_LAUNCH;
    }
    return DFINE_OK;
}
}
