// Error reporting + version of the C ABI (include/lafs_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include "common.hpp"
#include "lafs_hip.h"

static thread_local char g_err[512] = "";

extern "C" void lafs_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* lafs_last_error(void) { return g_err; }
extern "C" int lafs_version(void) { return LAFS_ABI_VERSION; }

// Diagnostic: what does ds_read_b64_tr_b16 return when lane l reads LDS bytes [8*l, 8*l+8) of a 512-element ramp?
// out[l*4 + j] = element index delivered to lane l, slot j.  Used by tests to pin the transpose-read model the
// attention / wgrad kernels are built on.
__global__ void debug_tr16_kernel(const short* in, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[512];
  const int l = threadIdx.x;
  for (int i = l; i < 512; i += 64) lds[i] = in[i];
  __syncthreads();
  s16x4_t v = lds_read_tr16(lds + l * 4);
  out[l * 4 + 0] = v[0]; out[l * 4 + 1] = v[1]; out[l * 4 + 2] = v[2]; out[l * 4 + 3] = v[3];
}
extern "C" int lafs_debug_tr16(const void* in, void* out, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  hipLaunchKernelGGL(debug_tr16_kernel, dim3(1), dim3(64), 0, stream, (const short*)in, (short*)out);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}
