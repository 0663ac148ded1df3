// HBM streaming probe: what do read-only, write-only, copy and read-2/write-1 kernels sustain on buffers far larger than the
// 256 MB Infinity Cache, as a function of workgroups per CU and 16-byte requests in flight per lane?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: read (sum), 1: write, 2: copy, 3: read two streams + write one (residual epilogue shape), 4: copy with non-temporal stores
template <int MODE, int U>
__global__ __launch_bounds__(256) void stream_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ c,
                                                     size_t n16, unsigned* sink) {
  const size_t stride = (size_t)gridDim.x * 256;
  u32x4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n16; i += U * stride) {
    u32x4 v[U], w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (MODE != 1) v[u] = a[i + u * stride];
      if (MODE == 3) w[u] = b[i + u * stride];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (MODE == 0) acc += v[u];
      else if (MODE == 1) c[i + u * stride] = u32x4{(unsigned)i, 1, 2, 3};
      else if (MODE == 2) c[i + u * stride] = v[u];
      else if (MODE == 3) c[i + u * stride] = v[u] + w[u];
      else __builtin_nontemporal_store(v[u], c + i + u * stride);
    }
  }
  if (MODE == 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *sink = 1;
}

template <int MODE, int U>
void run(const char* name, u32x4* a, u32x4* b, u32x4* c, size_t n16, unsigned* sink) {
  const double bytes = (double)n16 * 16 * (MODE == 0 || MODE == 1 ? 1 : (MODE == 3 ? 3 : 2));
  for (int per_cu : {2, 4, 8}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * per_cu;
    hipLaunchKernelGGL((stream_kernel<MODE, U>), dim3(grid), dim3(256), 0, 0, a, b, c, n16, sink);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((stream_kernel<MODE, U>), dim3(grid), dim3(256), 0, 0, a, b, c, n16, sink);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s U=%d wg/CU=%d  %7.1f us  %6.2f TB/s\n", name, U, per_cu, ms / 3 * 1e3, bytes / (ms / 3 * 1e-3) / 1e12);
  }
}

// Per-CU rates: the same streams from only `grid` workgroups (one or two per CU on grid <= 512): is a store stream bound by the
// CU's own memory pipeline or by the memory side?
template <int MODE, int U>
void run_cus(const char* name, u32x4* a, u32x4* b, u32x4* c, size_t n16, unsigned* sink) {
  const double bytes = (double)n16 * 16 * (MODE == 0 || MODE == 1 ? 1 : (MODE == 3 ? 3 : 2));
  for (int grid : {16, 32, 64, 128, 256, 512}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((stream_kernel<MODE, U>), dim3(grid), dim3(256), 0, 0, a, b, c, n16, sink);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((stream_kernel<MODE, U>), dim3(grid), dim3(256), 0, 0, a, b, c, n16, sink);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tbs = bytes / (ms * 1e-3) / 1e12;
    printf("%-28s U=%d workgroups=%3d  %8.1f us  %6.2f TB/s  = %6.1f GB/s per workgroup\n", name, U, grid, ms * 1e3, tbs, tbs * 1e3 / grid);
  }
}

// GEMM-epilogue-shaped writes: workgroup b owns a [TR rows] x [SEG bytes] tile of a row-major [rows][ROWB bytes] tensor (and the
// same tile of a second tensor when TWO), tiles numbered n-fastest, XCD b%8 walking a contiguous run of tiles like the GEMMs do.
template <int TR, int SEG, bool TWO>
__global__ __launch_bounds__(256) void tile_write_kernel(unsigned char* __restrict__ c, unsigned char* __restrict__ c2, int rows, int rowb) {
  const int tiles_n = rowb / SEG, n_tiles = gridDim.x;
  const int q = n_tiles >> 3, r = n_tiles & 7, x = blockIdx.x & 7, i = blockIdx.x >> 3;
  const int tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  constexpr int PER_ROW = SEG / 16;                    // 16-byte pieces per row segment
  for (int e = threadIdx.x; e < TR * PER_ROW; e += 256) {
    const int row = tm * TR + e / PER_ROW, piece = e % PER_ROW;
    if (row >= rows) continue;
    const size_t off = (size_t)row * rowb + (size_t)tn * SEG + piece * 16;
    *reinterpret_cast<u32x4*>(c + off) = u32x4{(unsigned)e, 1, 2, 3};
    if (TWO) *reinterpret_cast<u32x4*>(c2 + off) = u32x4{(unsigned)e, 5, 6, 7};
  }
}
template <int TR, int SEG, bool TWO>
void run_tiles(const char* name, unsigned char* c, unsigned char* c2, int rows, int rowb) {
  const int grid = ((rows + TR - 1) / TR) * (rowb / SEG);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((tile_write_kernel<TR, SEG, TWO>), dim3(grid), dim3(256), 0, 0, c, c2, rows, rowb);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((tile_write_kernel<TR, SEG, TWO>), dim3(grid), dim3(256), 0, 0, c, c2, rows, rowb);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)rows * rowb * (TWO ? 2 : 1);
  printf("%-44s tile %3d rows x %4d B%s: %7.1f us  %6.2f TB/s\n", name, TR, SEG, TWO ? " x2 tensors" : "", ms / 3 * 1e3, bytes / (ms / 3 * 1e-3) / 1e12);
}

int main() {
  const size_t n16 = (size_t)1 << 26;                  // 1 GiB per buffer
  u32x4 *a, *b, *c; unsigned* sink;
  hipMalloc(&a, n16 * 16); hipMalloc(&b, n16 * 16); hipMalloc(&c, n16 * 16); hipMalloc(&sink, 4);
  hipMemset(a, 1, n16 * 16); hipMemset(b, 2, n16 * 16); hipMemset(c, 0, n16 * 16);
  run<0, 1>("read", a, b, c, n16, sink); run<0, 4>("read", a, b, c, n16, sink); run<0, 8>("read", a, b, c, n16, sink);
  run<1, 1>("write", a, b, c, n16, sink); run<1, 4>("write", a, b, c, n16, sink);
  run<2, 1>("copy", a, b, c, n16, sink); run<2, 4>("copy", a, b, c, n16, sink); run<2, 8>("copy", a, b, c, n16, sink);
  run<4, 4>("copy, non-temporal stores", a, b, c, n16, sink);
  run<3, 4>("read 2 + write 1", a, b, c, n16, sink);
  if (getenv("LAB_PER_CU")) {
    const size_t n = (size_t)1 << 24;                  // 256 MiB
    run_cus<1, 4>("write", a, b, c, n, sink); run_cus<0, 4>("read", a, b, c, n, sink); run_cus<2, 4>("copy", a, b, c, n, sink);
    run_cus<0, 4>("read 32 MiB (cache-resident)", a, b, c, (size_t)1 << 21, sink);
    return 0;
  }
  // working sets that fit the Infinity Cache: 64 MiB read after write
  const size_t small = (size_t)1 << 22;
  run<2, 4>("copy 64 MiB (cache-resident)", a, b, c, small, sink);
  run<0, 4>("read 64 MiB (cache-resident)", a, b, c, small, sink);
  // the fc1 forward's outputs: two [44160][1536] bf16 tensors
  unsigned char* cc = (unsigned char*)c; unsigned char* cc2 = cc + ((size_t)512 << 20);
  run_tiles<256, 256, true>("fc1 outputs, 256x128 tiles", cc, cc2, 44160, 3072);
  run_tiles<128, 256, true>("fc1 outputs, 128x128 tiles", cc, cc2, 44160, 3072);
  run_tiles<256, 512, true>("fc1 outputs, 256x256 tiles", cc, cc2, 44160, 3072);
  run_tiles<128, 1024, true>("fc1 outputs, 128x512 tiles", cc, cc2, 44160, 3072);
  run_tiles<64, 3072, true>("fc1 outputs, 64 full rows", cc, cc2, 44160, 3072);
  run_tiles<256, 256, false>("one tensor, 256x128 tiles", cc, cc2, 44160, 3072);
  run_tiles<64, 3072, false>("one tensor, 64 full rows", cc, cc2, 44160, 3072);
  run_tiles<64, 3072, false>("2x rows one tensor, 64 full rows", cc, cc2, 88320, 3072);
  hipDeviceSynchronize();
  return 0;
}
