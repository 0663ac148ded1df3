// lafs_ctx: the per-device handle of the C ABI ("library holds no global state except a lazily created
// per-device handle, re-entrant per stream").  The reference has no counterpart -- its modules run every op on torch's current
// stream (lafs_train.py:513-613); the handle exists because THIS library forks a trunk pass over side streams of its own.
// Owner: the caller (one per engine in lafs_cvpr2024_amd/engine.py / finetune_engine.py); two contexts share nothing, so two
// engines driven from two host threads on two caller streams do not meet in any stream, event or switch.
#include <new>
#include "common.hpp"
#include "ctx.hpp"

static const int kDefaults[LAFS_OPT_COUNT] = {
    /* SIDE_STREAMS */ 1, /* ROW_CHAINS */ 2, /* KRES_MASK */ 15, /* KRES_MIN_ITEMS */ 4, /* NT_WIDE */ 1, /* NT_TALL */ 1, /* COMM_CUS */ 0,
    /* NT_BIG */ 1, /* MLP_FUSED */ 79};

int lafs_ctx_opt(const lafs_ctx* c, int opt) {
  if (opt < 0 || opt >= LAFS_OPT_COUNT) return 0;
  if (c == nullptr) return opt == LAFS_OPT_SIDE_STREAMS ? 0 : kDefaults[opt];
  return c->opt[opt];
}

extern "C" lafs_ctx* lafs_ctx_create(int device) {
  LAFS_CLEAR_ERROR();
  int prev = -1;
  if (hipGetDevice(&prev) != hipSuccess) { lafs_set_error("lafs_ctx_create: no HIP device"); return nullptr; }
  if (device < 0) device = prev;
  lafs_ctx* c = new (std::nothrow) lafs_ctx();
  if (c == nullptr) return nullptr;
  c->device = device;
  for (int i = 0; i < LAFS_OPT_COUNT; ++i) c->opt[i] = kDefaults[i];
  bool ok = (device == prev) || hipSetDevice(device) == hipSuccess;
  if (ok && hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) c->n_cu = 0;
  // streams and events exist from here on: never created lazily inside a hipGraph capture
  for (int i = 0; ok && i < 3; ++i) {
    ok = hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking) == hipSuccess &&
         hipEventCreateWithFlags(&c->join[i], hipEventDisableTiming) == hipSuccess;
  }
  ok = ok && hipEventCreateWithFlags(&c->fork, hipEventDisableTiming) == hipSuccess;
  for (int i = 0; ok && i < 64; ++i) {
    hipEvent_t e;
    ok = hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    if (ok) c->pool.push_back(e);
  }
  if (device != prev) (void)hipSetDevice(prev);
  c->streams_ok = ok;
  if (!ok) {
    lafs_set_error("lafs_ctx_create: could not create the side streams / events on device %d", device);
    lafs_ctx_destroy(c);
    return nullptr;
  }
  return c;
}

extern "C" void lafs_ctx_destroy(lafs_ctx* c) {
  if (c == nullptr) return;
  for (hipEvent_t e : c->pool) (void)hipEventDestroy(e);
  if (c->fork) (void)hipEventDestroy(c->fork);
  for (int i = 0; i < 3; ++i) {
    if (c->join[i]) (void)hipEventDestroy(c->join[i]);
    if (c->side[i]) (void)hipStreamDestroy(c->side[i]);
  }
  delete c;
}

extern "C" int lafs_ctx_set(lafs_ctx* c, int opt, int value) {
  if (c == nullptr || opt < 0 || opt >= LAFS_OPT_COUNT) return LAFS_EINVAL;
  if (opt == LAFS_OPT_COMM_CUS) value = value < 0 ? 0 : (value > 192 ? 192 : value);
  if (opt == LAFS_OPT_ROW_CHAINS) value = (value >= 4) ? 4 : (value >= 2 ? 2 : 1);
  c->opt[opt] = value;
  return LAFS_OK;
}

extern "C" int lafs_ctx_get(const lafs_ctx* c, int opt) { return lafs_ctx_opt(c, opt); }
