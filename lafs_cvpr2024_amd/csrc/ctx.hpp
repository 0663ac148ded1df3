// Internal layout of lafs_ctx (include/lafs_hip.h): the per-device handle that owns every device object the library itself creates
// -- the side streams and fork / join events of the trunk passes' row chains, the event pool of the two-stream backward -- and the
// kernel-selection options.  Nothing in the library is process-global any more: no function-local statics, no getenv.
#pragma once
#include <vector>
#include <hip/hip_runtime.h>
#include "lafs_hip.h"

struct lafs_ctx {
  int device = 0;
  int n_cu = 0;                                             // compute units of the device (one-workgroup-per-CU kernels size their rounds by it)
  hipStream_t side[3] = {nullptr, nullptr, nullptr};       // second .. fourth row chain / second attention group
  hipEvent_t fork = nullptr, join[3] = {nullptr, nullptr, nullptr};
  std::vector<hipEvent_t> pool;                             // weight-gradient stream protocol (lafs_trunk_backward)
  int opt[LAFS_OPT_COUNT];
  bool streams_ok = false;
};

// option value of a (possibly NULL) context: NULL means the compiled-in defaults with no side streams
int lafs_ctx_opt(const lafs_ctx* c, int opt);
