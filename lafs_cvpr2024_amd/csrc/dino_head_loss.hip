// DINO head's last layer + DINO loss in one piece: the [ncrops B + 2 B, K] logits never reach HBM.
//
// Replaces, for the training step, the last nn.Linear of DINOHead (weight-normalised, bias-free; vision_transformer.py:295-301) of
// BOTH networks followed by DINOLoss.forward and the row sums of update_center (lafs_train.py:643-679): the reference writes the
// fp32 logits of student and teacher (307 MB at batch 64, K = 100 000), reads them for 18 log_softmax passes and again for the
// centre.  The unfused kernels of this library (lafs_gemm_nt + lafs_dino_loss_fwd_bwd + lafs_colsum_f32) still wrote them once and
// read them twice: 0.9 GB of the step's 59.5 GB and 0.32 ms of its critical path (profiles/round4_serial_kernel_table.txt).
// Here the logits of a 64-class block are a K = 256 contraction away -- 2 x 100 000 x 256 bf16 of normalised weights is all that
// streams -- so they are formed twice instead of stored:
//   pass 1 (head_stats):  per 64-class block: T = zn_t Wn_t^T (teacher rows), S = zn_s Wn_s^T (student rows) on
//                         v_mfma_f32_16x16x32_bf16, per row the block's (max, sum exp) of (t - c) / tau_t resp. s / tau_s in the
//                         log2 domain, and the block's column sums of the raw teacher logits (the centre update);
//           (row_lse):    per row the blocks' (max, sum) pairs folded into a log-sum-exp, in block order;
//   pass 2 (head_grad):   the same tiles again: q_0, q_1 of the block into LDS, then for every student row
//                         p = exp(s / tau_s - lse), dL/ds = coef (n_v p - sum_{i != v} q_i) as bf16 (the operand of the two
//                         last-layer gradient GEMMs), and the block's share of sum_{v, b} <q, s / tau_s>;
//           (finalize):   loss = 1 / (n_terms B) (sum_{v, b} n_v lse_vb - sum of the blocks' shares), in block order.
// Same arithmetic as the unfused path: the MFMA k order per logit is that of the tiled kernel's 32-deep stages, the softmax
// statistics are folded in a fixed order (no atomics anywhere): results are run-to-run deterministic.
#include "common.hpp"
#include "lafs_hip.h"

namespace {

constexpr int CB = 64;                   // classes per workgroup (two workgroups per CU: 68 KB of LDS each in pass 2)
constexpr int DB = 256;                  // bottleneck width (reduction length)
constexpr int WROW = DB * 2;             // bytes per weight row in LDS
constexpr int NKK = DB / 32;             // k steps
constexpr int MAXTR = 128;               // teacher rows (2 B) the q tile holds

struct HArgs {
  const bf16_t* zs; const bf16_t* zt; const bf16_t* ws; const bf16_t* wt;      // [rows, 256], [Kpad, 256]
  const float* center;
  int ncrops, B, K, Kpad, nblk;
  const float* temps;                    // device {student_temp, teacher_temp}
  float its, itt;                        // host values when temps == nullptr
  float* part;                           // [rows_s + rows_t][nblk][2]: (max, sum) per row and class block, log2 domain
  float* stats;                          // [rows_s + rows_t]: log2-domain log-sum-exp
  float* colsum;                         // [K]: sum over the teacher rows of the raw logits
  bf16_t* grad; int ldg; float coef;
  float* dots;                           // [nblk]
  float* loss;
};

// MFMA row slot rho (column group j = rho >> 4, lane group g = (rho >> 2) & 3, register r = rho & 3) takes the class that makes a
// lane's registers of two column groups 8 CONSECUTIVE classes (gemm.hip: nt_perm): class = (j >> 1) * 32 + g * 8 + (j & 1) * 4 + r
__device__ __forceinline__ int cls_of_slot(int rho) {
  const int j = rho >> 4, g = (rho >> 2) & 3, r = rho & 3;
  return (j >> 1) * 32 + g * 8 + (j & 1) * 4 + r;
}
// class of this lane's register r of column group j (relative to the block)
__device__ __forceinline__ int lane_cls(int j, int fq, int r) { return (j >> 1) * 32 + fq * 8 + (j & 1) * 4 + r; }

// weights of class block `cb` of `w` into LDS: row rho (an MFMA row slot) = 512 B, 16-byte chunk c at chunk position c ^ (rho & 7)
__device__ __forceinline__ void stage_weights(unsigned char* lds, const bf16_t* w, int cb, int Kpad) {
  for (int i = threadIdx.x; i < CB * 32; i += 256) {
    const int rho = i >> 5, c = i & 31;
    const int cls = min(cb * CB + cls_of_slot(rho), Kpad - 1);
    const uint4 v = *reinterpret_cast<const uint4*>(w + (size_t)cls * DB + c * 8);
    *reinterpret_cast<uint4*>(lds + rho * WROW + ((c ^ (rho & 7)) << 4)) = v;
  }
}
// logits of 32 rows x 64 classes: acc[h][j][r] = class lane_cls(j, fq, r) of row 16 h + t16 (C^T blocks: MFMA rows = classes); every
// weight fragment read feeds two MFMAs
__device__ __forceinline__ void tile32(const unsigned char* wl, const bf16_t* z, int row, int rows, f32x4_t (&acc)[2][4]) {
  const int lane = threadIdx.x & 63, t16 = lane & 15, fq = lane >> 4;
  bf16x8_t zf[2][NKK];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const bf16_t* zr = z + (size_t)min(row + 16 * h + t16, rows - 1) * DB + fq * 8;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) zf[h][kk] = *reinterpret_cast<const bf16x8_t*>(zr + kk * 32);
  }
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[h][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rho = j * 16 + t16;
      const bf16x8_t wf = *reinterpret_cast<const bf16x8_t*>(wl + rho * WROW + (((kk * 4 + fq) ^ (rho & 7)) << 4));
      acc[0][j] = mfma16(wf, zf[0][kk], acc[0][j]);
      acc[1][j] = mfma16(wf, zf[1][kk], acc[1][j]);
    }
    // (left alone, hipcc hoists all 32 fragment reads of the tile above its MFMAs: 128 registers, spills at two workgroups per CU;
    // the other seven waves of the CU cover a k step's LDS latency)
    if (kk & 1) __builtin_amdgcn_sched_barrier(0);
  }
}

// ---------------------------------------------------------------------------------------------- pass 1
// (max, sum exp2) of one row's 64 values of this block (spread over the 4 lane groups fq), written by lane group 0
__device__ __forceinline__ void row_part(f32x4_t (&a)[4], float* out, bool write) {
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) m = fmaxf(m, a[j][r]);
  m = fmaxf(m, __shfl_xor(m, 16, 64)); m = fmaxf(m, __shfl_xor(m, 32, 64));
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) s += __builtin_amdgcn_exp2f(a[j][r] - m);       // (a block has at least one class < K: m is finite)
  s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
  if (write) { out[0] = m; out[1] = s; }
}

__global__ __launch_bounds__(256, 2) void head_stats_kernel(HArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char wl[CB * WROW];
  __shared__ float csum[4][CB];
  const int cb = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, t16 = lane & 15, fq = lane >> 4;
  const float its = p.temps ? 1.0f / p.temps[0] : p.its, itt = p.temps ? 1.0f / p.temps[1] : p.itt;
  const float cs2 = its * 1.4426950408889634f, ct2 = itt * 1.4426950408889634f;
  const int rows_s = p.ncrops * p.B, rows_t = 2 * p.B;
  const int n0 = cb * CB;
  // ---- teacher rows: statistics of (t - c) / tau_t, column sums of t
  stage_weights(wl, p.wt, cb, p.Kpad);
  __syncthreads();
  float cen[4][4], col[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + lane_cls(j, fq, r);
      cen[j][r] = n < p.K ? p.center[n] : 0.f;
      col[j][r] = 0.f;
    }
#pragma unroll 1
  for (int rb = wave; rb * 32 < rows_t; rb += 4) {
    f32x4_t acc[2][4];
    tile32(wl, p.zt, rb * 32, rows_t, acc);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = rb * 32 + 16 * h + t16;
      const bool rok = row < rows_t;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = n0 + lane_cls(j, fq, r) < p.K;
          if (rok && ok) col[j][r] += acc[h][j][r];
          acc[h][j][r] = ok ? (acc[h][j][r] - cen[j][r]) * ct2 : -INFINITY;
        }
      row_part(acc[h], p.part + ((size_t)(rows_s + min(row, rows_t - 1)) * p.nblk + cb) * 2, fq == 0 && rok);
    }
  }
  // column sums: over this wave's rows (lanes t16), then over the waves
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = col[j][r];
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
      if (t16 == 0) csum[wave][lane_cls(j, fq, r)] = v;
    }
  __syncthreads();
  if (threadIdx.x < CB && n0 + threadIdx.x < p.K && p.colsum != nullptr)
    p.colsum[n0 + threadIdx.x] = (csum[0][threadIdx.x] + csum[1][threadIdx.x]) + (csum[2][threadIdx.x] + csum[3][threadIdx.x]);
  __syncthreads();
  // ---- student rows: statistics of s / tau_s
  stage_weights(wl, p.ws, cb, p.Kpad);
  __syncthreads();
#pragma unroll 1
  for (int rb = wave; rb * 32 < rows_s; rb += 4) {
    f32x4_t acc[2][4];
    tile32(wl, p.zs, rb * 32, rows_s, acc);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = rb * 32 + 16 * h + t16;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[h][j][r] = (n0 + lane_cls(j, fq, r) < p.K) ? acc[h][j][r] * cs2 : -INFINITY;
      row_part(acc[h], p.part + ((size_t)min(row, rows_s - 1) * p.nblk + cb) * 2, fq == 0 && row < rows_s);
    }
  }
}

// stats[row] = log2-domain log-sum-exp of the row: the blocks' (max, sum) pairs folded in block order per lane, then across the
// lanes of a wave and across the four waves in a fixed butterfly / wave order (one wave per row: four rows per workgroup)
__global__ __launch_bounds__(256) void row_lse_kernel(HArgs p, int rows) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float2* part = reinterpret_cast<const float2*>(p.part + (size_t)row * p.nblk * 2);
  float m = -INFINITY, s = 0.f;
  for (int b = lane; b < p.nblk; b += 64) {
    const float2 v = part[b];
    const float mn = fmaxf(m, v.x);
    s = s * __builtin_amdgcn_exp2f(m - mn) + v.y * __builtin_amdgcn_exp2f(v.x - mn);
    m = mn;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float mo = __shfl_xor(m, o, 64), so = __shfl_xor(s, o, 64);
    const float mn = fmaxf(m, mo);
    s = (mn == -INFINITY) ? 0.f : s * __builtin_amdgcn_exp2f(m - mn) + so * __builtin_amdgcn_exp2f(mo - mn);
    m = mn;
  }
  if (lane == 0) p.stats[row] = m + __log2f(s);
}

// ---------------------------------------------------------------------------------------------- pass 2
__global__ __launch_bounds__(256, 2) void head_grad_kernel(HArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char wl[CB * WROW];
  __shared__ __attribute__((aligned(16))) float q[MAXTR][CB + 4];       // q_0 rows 0..B-1, q_1 rows B..2B-1 (+4: row stride off the bank period)
  __shared__ float dred[4];
  const int cb = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, t16 = lane & 15, fq = lane >> 4;
  const float its = p.temps ? 1.0f / p.temps[0] : p.its, itt = p.temps ? 1.0f / p.temps[1] : p.itt;
  const float cs2 = its * 1.4426950408889634f, ct2 = itt * 1.4426950408889634f;
  const float coef = p.temps ? p.coef / (p.its * p.temps[0]) : p.coef;       // (coef was formed with the host's student temperature)
  const int B = p.B, rows_s = p.ncrops * B, rows_t = 2 * B;
  const int n0 = cb * CB;
  if (n0 >= p.K) {                                       // a block of pad columns only (K .. Kpad): the gradient there is zero
    for (int i = threadIdx.x; i < rows_s * (CB / 8); i += 256) {
      const int row = i / (CB / 8), c = n0 + (i % (CB / 8)) * 8;
      if (c + 8 <= p.Kpad) *reinterpret_cast<uint4*>(p.grad + (size_t)row * p.ldg + c) = make_uint4(0, 0, 0, 0);
    }
    return;
  }
  stage_weights(wl, p.wt, cb, p.Kpad);
  __syncthreads();
#pragma unroll 1
  for (int rb = wave; rb * 32 < rows_t; rb += 4) {
    f32x4_t acc[2][4];
    tile32(wl, p.zt, rb * 32, rows_t, acc);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = rb * 32 + 16 * h + t16;
      if (row < rows_t) {
        const float l = p.stats[rows_s + row];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float4 v;
          float* pv = &v.x;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = n0 + lane_cls(j, fq, r);
            pv[r] = n < p.K ? __builtin_amdgcn_exp2f((acc[h][j][r] - p.center[n]) * ct2 - l) : 0.f;
          }
          *reinterpret_cast<float4*>(&q[row][lane_cls(j, fq, 0)]) = v;
        }
      }
    }
  }
  __syncthreads();
  stage_weights(wl, p.ws, cb, p.Kpad);
  __syncthreads();
  float dot = 0.f;
#pragma unroll 1
  for (int rb = wave; rb * 32 < rows_s; rb += 4) {
    f32x4_t acc[2][4];
    tile32(wl, p.zs, rb * 32, rows_s, acc);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = rb * 32 + 16 * h + t16;
      if (row < rows_s) {
        const int v = row / B, b = row - v * B;
        const float l = p.stats[row];
        const float nv = v < 2 ? 1.f : 2.f;
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {                           // 8 consecutive classes: column groups 2 qq and 2 qq + 1
          const int c0 = qq * 32 + fq * 8;
          float g[8];
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const float4 q0 = *reinterpret_cast<const float4*>(&q[b][c0 + 4 * jj]);
            const float4 q1 = *reinterpret_cast<const float4*>(&q[B + b][c0 + 4 * jj]);
            const float qa[4] = {q0.x, q0.y, q0.z, q0.w}, qb[4] = {q1.x, q1.y, q1.z, q1.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const bool ok = n0 + c0 + 4 * jj + r < p.K;
              const float a = acc[h][2 * qq + jj][r];
              const float qs = v == 0 ? qb[r] : (v == 1 ? qa[r] : qa[r] + qb[r]);
              dot += ok ? qs * (a * its) : 0.f;
              g[4 * jj + r] = ok ? coef * (nv * __builtin_amdgcn_exp2f(a * cs2 - l) - qs) : 0.f;
            }
          }
          if (n0 + c0 + 8 <= p.Kpad)                               // (Kpad % 8 == 0: a piece lies inside the padded row or beyond it)
            *reinterpret_cast<uint4*>(p.grad + (size_t)row * p.ldg + n0 + c0) =
                make_uint4(pack_bf2(g[0], g[1]), pack_bf2(g[2], g[3]), pack_bf2(g[4], g[5]), pack_bf2(g[6], g[7]));
        }
      }
    }
  }
  dot = wave_sum(dot);
  if (lane == 0) dred[wave] = dot;
  __syncthreads();
  if (threadIdx.x == 0) p.dots[cb] = (dred[0] + dred[1]) + (dred[2] + dred[3]);
}

// loss = 1 / (n_terms B) * (sum_{v, b} n_v lse_vb - sum_blocks dots): one workgroup, fixed order
__global__ __launch_bounds__(256) void head_loss_final_kernel(HArgs p) {
  __shared__ float red[256];
  const int rows_s = p.ncrops * p.B;
  float acc = 0.f;
  for (int i = threadIdx.x; i < rows_s; i += 256) acc += ((i / p.B) < 2 ? 1.f : 2.f) * p.stats[i] * 0.6931471805599453f;     // log2 -> natural
  for (int i = threadIdx.x; i < p.nblk; i += 256) acc -= p.dots[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < 256; ++i) t += red[i];
    p.loss[0] = t / (float)((2 * p.ncrops - 2) * p.B);
  }
}

}  // namespace

extern "C" int64_t lafs_dino_head_loss_workspace(int ncrops, int B, int K) {
  if (ncrops < 2 || B <= 0 || K <= 0) return -1;
  const int64_t nblk = (K + CB - 1) / CB, rows = (int64_t)(ncrops + 2) * B;
  return rows * nblk * 2 + rows + nblk;
}

extern "C" int lafs_dino_head_loss(const void* zn_student, const void* zn_teacher, const void* wn_student, const void* wn_teacher,
                                   int dim, const float* center, int ncrops, int B, int K, int Kpad, float student_temp,
                                   float teacher_temp, const float* dev_temps, float* loss_out, void* grad_bf16, int ldg,
                                   float grad_scale, float* colsum_out, float* workspace, hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(zn_student && zn_teacher && wn_student && wn_teacher && center && loss_out && grad_bf16 && workspace, "null operand");
  LAFS_CHECK_ARG(dim == DB, "the fused head takes the DINOHead's default bottleneck width 256");
  LAFS_CHECK_ARG(ncrops >= 2 && ncrops <= 16 && B > 0 && 2 * B <= MAXTR, "2 <= ncrops <= 16, batch <= 64 per rank");
  LAFS_CHECK_ARG(K > 0 && Kpad >= K && Kpad % 8 == 0 && ldg % 8 == 0 && ldg >= Kpad, "K <= Kpad <= ldg, multiples of 8");
  HArgs a = {};
  a.zs = (const bf16_t*)zn_student; a.zt = (const bf16_t*)zn_teacher; a.ws = (const bf16_t*)wn_student; a.wt = (const bf16_t*)wn_teacher;
  a.center = center; a.ncrops = ncrops; a.B = B; a.K = K; a.Kpad = Kpad; a.nblk = ceil_div(K, CB);
  a.temps = dev_temps; a.its = 1.0f / student_temp; a.itt = 1.0f / teacher_temp;
  const int64_t rows = (int64_t)(ncrops + 2) * B;
  a.part = workspace; a.stats = a.part + rows * a.nblk * 2; a.dots = a.stats + rows;
  a.colsum = colsum_out; a.grad = (bf16_t*)grad_bf16; a.ldg = ldg;
  a.coef = grad_scale / ((float)(2 * ncrops - 2) * (float)B * student_temp);
  a.loss = loss_out;
  hipLaunchKernelGGL(head_stats_kernel, dim3(a.nblk), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(row_lse_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, a, (int)rows);
  hipLaunchKernelGGL(head_grad_kernel, dim3(ceil_div(Kpad, CB)), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(head_loss_final_kernel, dim3(1), dim3(256), 0, stream, a);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}
