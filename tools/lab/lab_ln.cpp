// LayerNorm streaming probe: the product's one-row-per-wave kernels (csrc/layernorm.hip) against forms that keep more rows in flight per
// wave (LPR lanes per row: 64 = one row per wave, 32 = two rows per wave), on buffer sets rotated through more than the 256 MB Infinity
// Cache.  Prints us and TB/s of algorithmic bytes (bwd: x f32 + dy bf16 + g f32 read, g f32 + gb bf16 written = 14 B per element;
// fwd: 4 read + 2 written).   make -C tools/lab lab_ln && gpurun -- tools/lab/lab_ln
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef unsigned short bf16_t;
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  uint32_t a = __float_as_uint(lo), b = __float_as_uint(hi);
  a += 0x7fffu + ((a >> 16) & 1u); b += 0x7fffu + ((b >> 16) & 1u);
  return (a >> 16) | (b & 0xffff0000u);
}
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
template <int LPR> __device__ __forceinline__ float row_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---------------------------------------------------------------- backward
// LPR lanes own a row: lane l (within the row group) holds float4 at columns l*4 + LPR*4*i, i < NI (D = NI * LPR * 4).
// PERS: 0 = grid covers the rows once (no loop, no prefetch), 1 = grid-stride with the next row group's operands requested a trip ahead.
template <int D, int LPR, int NW, int PERS, int ABL = 0>
__global__ __launch_bounds__(NW * 64) void ln_bwd(const bf16_t* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ stats,
                                                  const float* __restrict__ gamma, float* __restrict__ g_io, bf16_t* __restrict__ gb,
                                                  float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, float* __restrict__ g_out = nullptr) {
  constexpr int NI = (D + LPR * 4 - 1) / (LPR * 4), RPW = 64 / LPR;
  __shared__ float red[NW * RPW][D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, l = lane % LPR;
  auto on = [&](int i) { return (i + 1) * LPR * 4 <= D || l * 4 + LPR * 4 * i < D; };   // (compile-time true for all but a ragged last chunk)
  float4 gam[NI], ag[NI], ab[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    gam[i] = on(i) ? *reinterpret_cast<const float4*>(gamma + l * 4 + LPR * 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
    ag[i] = make_float4(0.f, 0.f, 0.f, 0.f); ab[i] = ag[i];
  }
  struct RowIn { float4 xv[NI] = {}, old[NI] = {}; uint2 dw[NI] = {}; float mean, rstd; };
  auto load_row = [&](int row, RowIn& r) {
    if (ABL & 8) { r.mean = 0.f; r.rstd = 1.f; } else { r.mean = stats[2 * row]; r.rstd = stats[2 * row + 1]; }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = l * 4 + LPR * 4 * i;
      if (on(i)) {
        r.xv[i] = *reinterpret_cast<const float4*>(x + (size_t)row * D + c);
        r.dw[i] = *reinterpret_cast<const uint2*>(dy + (size_t)row * D + c);
        if (!(ABL & 16)) r.old[i] = *reinterpret_cast<const float4*>(g_io + (size_t)row * D + c);
      }
    }
  };
  auto work = [&](int row, const RowIn& cur) {
    const float mean = cur.mean, rstd = cur.rstd;
    float4 xh[NI], d[NI];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (!on(i)) { xh[i] = make_float4(0.f, 0.f, 0.f, 0.f); d[i] = xh[i]; continue; }
      const float4 xv = cur.xv[i];
      d[i] = make_float4(bf_lo(cur.dw[i].x), bf_hi(cur.dw[i].x), bf_lo(cur.dw[i].y), bf_hi(cur.dw[i].y));
      xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
      if (!(ABL & 1)) {
      ag[i].x += d[i].x * xh[i].x; ag[i].y += d[i].y * xh[i].y; ag[i].z += d[i].z * xh[i].z; ag[i].w += d[i].w * xh[i].w;
      ab[i].x += d[i].x; ab[i].y += d[i].y; ab[i].z += d[i].z; ab[i].w += d[i].w;
      }
      d[i].x *= gam[i].x; d[i].y *= gam[i].y; d[i].z *= gam[i].z; d[i].w *= gam[i].w;
      s1 += d[i].x + d[i].y + d[i].z + d[i].w;
      s2 += d[i].x * xh[i].x + d[i].y * xh[i].y + d[i].z * xh[i].z + d[i].w * xh[i].w;
    }
    const float m1 = row_sum<LPR>(s1) / (float)D, m2 = row_sum<LPR>(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = l * 4 + LPR * 4 * i;
      if (!on(i)) continue;
      float4 o = make_float4(rstd * (d[i].x - m1 - xh[i].x * m2), rstd * (d[i].y - m1 - xh[i].y * m2),
                             rstd * (d[i].z - m1 - xh[i].z * m2), rstd * (d[i].w - m1 - xh[i].w * m2));
      o.x += cur.old[i].x; o.y += cur.old[i].y; o.z += cur.old[i].z; o.w += cur.old[i].w;
      *reinterpret_cast<float4*>(((ABL & 2) ? g_out : g_io) + (size_t)row * D + c) = o;
      if (!(ABL & 4)) *reinterpret_cast<uint2*>(gb + (size_t)row * D + c) = make_uint2(pack_bf2(o.x, o.y), pack_bf2(o.z, o.w));
    }
  };
  const int stride = gridDim.x * NW * RPW;
  int row = (blockIdx.x * NW + wave) * RPW + sub;
  if (PERS == 0) {
    if (row < rows) { RowIn cur; load_row(row, cur); work(row, cur); }
  } else {
    RowIn cur;
    if (row < rows) load_row(row, cur);
    for (; row < rows; row += stride) {
      RowIn nxt;
      const bool more = row + stride < rows;
      if (more) load_row(row + stride, nxt);
      work(row, cur);
      if (more) cur = nxt;
    }
  }
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if (pass) __syncthreads();
#pragma unroll
    for (int i = 0; i < NI; ++i) if (on(i)) *reinterpret_cast<float4*>(&red[wave * RPW + sub][l * 4 + LPR * 4 * i]) = pass ? ab[i] : ag[i];
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += NW * 64) {
      float sg = 0.f;
#pragma unroll
      for (int w = 0; w < NW * RPW; ++w) sg += red[w][c];
      atomicAdd((pass ? dbeta : dgamma) + c, sg);
    }
  }
}

// ---------------------------------------------------------------- forward
template <int D, int LPR, int NW, int PERS>
__global__ __launch_bounds__(NW * 64) void ln_fwd(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  float eps, bf16_t* __restrict__ y, float* __restrict__ stats, int rows) {
  constexpr int NI = (D + LPR * 4 - 1) / (LPR * 4), RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, l = lane % LPR;
  auto on = [&](int i) { return (i + 1) * LPR * 4 <= D || l * 4 + LPR * 4 * i < D; };   // (compile-time true for all but a ragged last chunk)
  float4 g4[NI], b4[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    g4[i] = on(i) ? *reinterpret_cast<const float4*>(gamma + l * 4 + LPR * 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
    b4[i] = on(i) ? *reinterpret_cast<const float4*>(beta + l * 4 + LPR * 4 * i) : g4[i];
  }
  struct RowIn { float4 v[NI]; };
  auto load_row = [&](int row, RowIn& r) {
#pragma unroll
    for (int i = 0; i < NI; ++i) r.v[i] = on(i) ? *reinterpret_cast<const float4*>(x + (size_t)row * D + l * 4 + LPR * 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto work = [&](int row, const RowIn& r) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) s += r.v[i].x + r.v[i].y + r.v[i].z + r.v[i].w;
    const float mean = row_sum<LPR>(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const float a = r.v[i].x - mean, b = r.v[i].y - mean, c = r.v[i].z - mean, d = r.v[i].w - mean;
      if (on(i)) q += a * a + b * b + c * c + d * d;
    }
    const float rstd = rsqrtf(row_sum<LPR>(q) / (float)D + eps);
    if (l == 0) *reinterpret_cast<float2*>(stats + 2 * row) = make_float2(mean, rstd);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const float o0 = (r.v[i].x - mean) * rstd * g4[i].x + b4[i].x, o1 = (r.v[i].y - mean) * rstd * g4[i].y + b4[i].y;
      const float o2 = (r.v[i].z - mean) * rstd * g4[i].z + b4[i].z, o3 = (r.v[i].w - mean) * rstd * g4[i].w + b4[i].w;
      if (on(i)) *reinterpret_cast<uint2*>(y + (size_t)row * D + l * 4 + LPR * 4 * i) = make_uint2(pack_bf2(o0, o1), pack_bf2(o2, o3));
    }
  };
  const int stride = gridDim.x * NW * RPW;
  int row = (blockIdx.x * NW + wave) * RPW + sub;
  if (PERS == 0) {
    if (row < rows) { RowIn cur; load_row(row, cur); work(row, cur); }
  } else if (PERS == 1) {
    RowIn cur;
    if (row < rows) load_row(row, cur);
    for (; row < rows; row += stride) {
      RowIn nxt;
      const bool more = row + stride < rows;
      if (more) load_row(row + stride, nxt);
      work(row, cur);
      if (more) cur = nxt;
    }
  } else {                                              // PERS == 2: four row groups requested at once, no prefetch across trips
    for (; row < rows; row += 4 * stride) {
      RowIn r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) if (row + u * stride < rows) load_row(row + u * stride, r[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) if (row + u * stride < rows) work(row + u * stride, r[u]);
    }
  }
}

struct Set { float *x, *g, *stats; bf16_t *dy, *gb, *y; };

template <typename F>
float time_us(F launch, int reps) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) launch(r);
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) launch(r);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}

template <int D>
void run_shape(int rows, int nset) {
  std::vector<Set> sets(nset);
  const size_t n = (size_t)rows * D;
  std::vector<float> hx(n); std::vector<bf16_t> hd(n);
  unsigned s = 1;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; hx[i] = ((s >> 8) & 0xffff) / 65536.f - 0.5f; hd[i] = (bf16_t)(0x3c00 + ((s >> 4) & 0x1ff) + ((s >> 20) & 1) * 0x8000); }
  for (auto& t : sets) {
    (void)hipMalloc(&t.x, n * 4); (void)hipMalloc(&t.g, n * 4); (void)hipMalloc(&t.stats, (size_t)rows * 8);
    (void)hipMalloc(&t.dy, n * 2); (void)hipMalloc(&t.gb, n * 2); (void)hipMalloc(&t.y, n * 2);
    (void)hipMemcpy(t.x, hx.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(t.g, hx.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(t.dy, hd.data(), n * 2, hipMemcpyHostToDevice);
  }
  float *gamma, *beta, *dg, *db;
  (void)hipMalloc(&gamma, D * 4); (void)hipMalloc(&beta, D * 4); (void)hipMalloc(&dg, D * 4); (void)hipMalloc(&db, D * 4);
  std::vector<float> hg(D, 1.0f);
  (void)hipMemcpy(gamma, hg.data(), D * 4, hipMemcpyHostToDevice); (void)hipMemset(beta, 0, D * 4); (void)hipMemset(dg, 0, D * 4); (void)hipMemset(db, 0, D * 4);
  const int reps = 24;
  auto report = [&](const char* name, float us, double bytes_per_el) {
    printf("  D=%d rows=%5d %-58s %7.1f us  %5.2f TB/s\n", D, rows, name, us, bytes_per_el * n / (us * 1e-6) / 1e12);
  };
  // forward (writes the statistics the backward reads)
#define FWD(LPR, NW, PERS, GRID, label) \
  report("fwd " label, time_us([&](int r) { const Set& t = sets[r % nset]; hipLaunchKernelGGL((ln_fwd<D, LPR, NW, PERS>), dim3(GRID), dim3(NW * 64), 0, 0, t.x, gamma, beta, 1e-6f, t.y, t.stats, rows); }, reps), 6.0)
#define BWDA(ABL, BYTES, label) \
  report("bwd " label, time_us([&](int r) { const Set& t = sets[r % nset]; const Set& u = sets[(r + 1) % nset]; hipLaunchKernelGGL((ln_bwd<D, 64, (D == 384 ? 16 : 8), 1, ABL>), dim3(D == 384 ? 256 : 512), dim3((D == 384 ? 16 : 8) * 64), 0, 0, t.dy, t.x, t.stats, gamma, t.g, t.gb, dg, db, rows, u.g); }, reps), BYTES)
#define BWD(LPR, NW, PERS, GRID, label) \
  report("bwd " label, time_us([&](int r) { const Set& t = sets[r % nset]; hipLaunchKernelGGL((ln_bwd<D, LPR, NW, PERS>), dim3(GRID), dim3(NW * 64), 0, 0, t.dy, t.x, t.stats, gamma, t.g, t.gb, dg, db, rows); }, reps), 14.0)
  FWD(64, 4, 0, (rows + 3) / 4, "product: row per wave, 4-wave groups, one trip");
  FWD(64, 4, 1, 2048, "row per wave, grid-stride 2048 x 4 waves, prefetch 1");
  FWD(64, 4, 1, 1024, "row per wave, grid-stride 1024 x 4 waves, prefetch 1");
  FWD(64, 4, 2, 1024, "row per wave, 1024 x 4 waves, 4 rows at once");
  FWD(64, 4, 2, 512, "row per wave, 512 x 4 waves, 4 rows at once");
  FWD(32, 4, 0, (rows + 7) / 8, "two rows per wave, one trip");
  FWD(32, 4, 1, 1024, "two rows per wave, grid-stride 1024 x 4, prefetch 1");
  FWD(32, 4, 2, 512, "two rows per wave, 512 x 4 waves, 4 row pairs at once");
  FWD(32, 4, 2, 256, "two rows per wave, 256 x 4 waves, 4 row pairs at once");
  if constexpr (D == 384) {
    BWD(64, 16, 1, 256, "product: row per wave, 256 x 16 waves, prefetch 1");
    BWD(64, 8, 1, 512, "row per wave, 512 x 8 waves, prefetch 1");
    BWD(64, 4, 1, 1024, "row per wave, 1024 x 4 waves, prefetch 1");
    BWD(64, 4, 0, (rows + 3) / 4, "row per wave, one trip (atomics per 4 rows: timing only)");
    BWD(32, 16, 1, 256, "two rows per wave, 256 x 16 waves, prefetch 1");
    BWD(32, 8, 1, 512, "two rows per wave, 512 x 8 waves, prefetch 1");
    BWD(32, 8, 1, 256, "two rows per wave, 256 x 8 waves, prefetch 1");
    BWD(32, 4, 1, 1024, "two rows per wave, 1024 x 4 waves, prefetch 1");
    BWD(32, 4, 0, (rows + 7) / 8, "two rows per wave, one trip (atomics per 8 rows: timing only)");
  } else {
    BWD(64, 8, 1, 512, "product: row per wave, 512 x 8 waves, prefetch 1");
    BWD(64, 4, 1, 1024, "row per wave, 1024 x 4 waves, prefetch 1");
    BWD(64, 4, 1, 512, "row per wave, 512 x 4 waves, prefetch 1");
  }
  BWDA(0, 14.0, "ablation base (product form)");
  BWDA(1, 14.0, "no gamma / beta sums");
  BWDA(2, 14.0, "g written to another buffer (no read-modify-write)");
  BWDA(4, 12.0, "no bf16 copy store");
  BWDA(8, 14.0, "no statistics loads");
  BWDA(16, 10.0, "no load of the incoming g");
  BWDA(1 | 8, 14.0, "no sums, no statistics loads");
  BWDA(4 | 16, 8.0, "no bf16 store, no g load: x + dy in, g out");
  for (auto& t : sets) { (void)hipFree(t.x); (void)hipFree(t.g); (void)hipFree(t.stats); (void)hipFree(t.dy); (void)hipFree(t.gb); (void)hipFree(t.y); }
}

int main(int argc, char** argv) {
  const int nset = argc > 1 ? atoi(argv[1]) : 6;
  printf("buffer sets rotated: %d\n", nset);
  run_shape<384>(25216, nset);
  run_shape<384>(18944, nset);
  run_shape<768>(25216, nset);
  return 0;
}
