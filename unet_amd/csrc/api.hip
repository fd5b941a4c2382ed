// Error plumbing and version of the libunetk C ABI.
#include <stdarg.h>
#include <stdio.h>

#include <atomic>
#include <mutex>

#include "common.h"

static thread_local char g_err[512] = "";

void unetk_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Events for unetk_stream_fork: a small ring per device, created on first use and kept until the process ends
// (an event may be recorded again while an earlier wait on it is pending: the wait took the state at its call).
#define UNETK_CHECK_HIP(call)                                                  \
  do {                                                                         \
    hipError_t e_ = (call);                                                    \
    if (e_ != hipSuccess) {                                                    \
      unetk_set_error("%s failed: %s", #call, hipGetErrorString(e_));          \
      return UNETK_EHIP;                                                       \
    }                                                                          \
  } while (0)

namespace {
constexpr int FORK_EVENTS = 64, FORK_DEVICES = 16;
hipEvent_t g_fork_ev[FORK_DEVICES][FORK_EVENTS];
std::atomic<int> g_fork_made[FORK_DEVICES];
std::atomic<unsigned> g_fork_next[FORK_DEVICES];
std::mutex g_fork_mutex;
}  // namespace

extern "C" {
const char* unetk_last_error(void) { return g_err; }
int unetk_version(void) { return 100; }

int unetk_stream_fork(hipStream_t from, hipStream_t to) {
  int dev = 0;
  UNETK_CHECK_HIP(hipGetDevice(&dev));
  UNETK_CHECK_ARG(dev >= 0 && dev < FORK_DEVICES, "stream_fork: device %d", dev);
  if (g_fork_made[dev].load(std::memory_order_acquire) == 0) {
    std::lock_guard<std::mutex> lock(g_fork_mutex);
    if (g_fork_made[dev].load(std::memory_order_relaxed) == 0) {
      for (int i = 0; i < FORK_EVENTS; ++i) UNETK_CHECK_HIP(hipEventCreateWithFlags(&g_fork_ev[dev][i], hipEventDisableTiming));
      g_fork_made[dev].store(1, std::memory_order_release);
    }
  }
  hipEvent_t ev = g_fork_ev[dev][g_fork_next[dev].fetch_add(1, std::memory_order_relaxed) % FORK_EVENTS];
  UNETK_CHECK_HIP(hipEventRecord(ev, from));
  UNETK_CHECK_HIP(hipStreamWaitEvent(to, ev, 0));
  return UNETK_OK;
}
}
