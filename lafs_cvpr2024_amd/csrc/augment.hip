// Device-side view augmentation of the LAFS loader (reference lafs_train.py:790-886, DataAugmentation_LAFS): per (image, crop)
// one workgroup builds the 112x112 RandomResizedCrop (Pillow bicubic, fixed point, two passes) in LDS, writes the clean view,
// then runs ColorJitter / grayscale / GaussianBlur (Pillow's 3-pass box approximation) / solarize on the LDS copy and writes the
// augmented view.  Every arithmetic convention (8-bit intermediates, truncations, float vs double) follows Pillow's C code so
// that the result is bit-identical to the PIL pipeline torchvision drives (oracle/augment.py is pinned against Pillow; this
// kernel against the oracle).  The image never leaves the CU between the ten-odd PIL calls the reference makes per view.
#include "common.hpp"
#include "lafs_hip.h"

// Pillow's results depend on every float operation being rounded on its own: no fused multiply-add anywhere in this file
// (hipcc contracts a*b+c by default, and HIP's __fmul_rn / __fadd_rn are plain operators that contract as well).
#pragma STDC FP_CONTRACT OFF

namespace {

constexpr int S = 112, NPIX = S * S, NBYTE = NPIX * 3;
constexpr int PW = 20;                 // int32 words per parameter record (lafs_cvpr2024_amd/augment.py: pack_params)
constexpr int TW = 8;                  // table words per output index: first source index, taps, 6 coefficients
constexpr int PREC = 22;               // Resample.c PRECISION_BITS for 8-bit images

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ __forceinline__ int luma(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

// Image.blend(degenerate d, image v, alpha) (Blend.c): float arithmetic WITHOUT contraction, truncation, clip when extrapolating
__device__ __forceinline__ int blend(int d, int v, float a, bool interp) {
#pragma clang fp contract(off)
  const float prod = a * (float)(v - d);              // two separately rounded operations (plain operators under contract(off);
  const float t = (float)d + prod;                    // HIP's __fmul_rn/__fadd_rn are inlined WITH contraction allowed)
  if (interp) return (int)t;
  if (t <= 0.0f) return 0;
  if (t >= 255.0f) return 255;
  return (int)t;
}

__device__ void rgb2hsv(int r, int g, int b, int& uh, int& us, int& uv) {          // Convert.c rgb2hsv_row
#pragma clang fp contract(off)
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  uv = maxc;
  if (minc == maxc) { uh = 0; us = 0; return; }
  const float cr = (float)(maxc - minc);
  const float s = cr / (float)maxc;
  const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
  float h;
  if (r == maxc) h = bc - gc;
  else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
  else h = (float)(4.0 + (double)gc - (double)rc);
  h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
  uh = clip8((int)((double)h * 255.0));
  us = clip8((int)((double)s * 255.0));
}

__device__ void hsv2rgb(int h, int s, int v, int& r, int& g, int& b) {               // Convert.c hsv2rgb
#pragma clang fp contract(off)
  if (s == 0) { r = g = b = v; return; }
  const double fh = (double)h * 6.0 / 255.0;
  const int i = (int)floor(fh);
  const float f = (float)(fh - (double)i);
  const float fs = (float)s / 255.0f, vf = (float)v;
  const float one_fs = 1.0f - fs;
  const float fsf = fs * f, one_fsf = 1.0f - fsf;
  const float omf = 1.0f - f, fsomf = fs * omf, one_fsomf = 1.0f - fsomf;
  const float pf = vf * one_fs, qf = vf * one_fsf, tf = vf * one_fsomf;
  const int p = clip8((int)rint((double)pf));
  const int q = clip8((int)rint((double)qf));
  const int t = clip8((int)rint((double)tf));
  switch (i % 6) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}

__device__ __forceinline__ float norm_px(int v) {                                    // ToTensor + Normalize(0.5, 0.5)
#pragma clang fp contract(off)
  const float x = (float)v / 255.0f;
  const float y = x - 0.5f;
  return y / 0.5f;
}

__device__ void write_view(const unsigned char* img, float* __restrict__ out) {      // HWC uint8 in LDS -> CHW float in HBM
  for (int e = threadIdx.x; e < NBYTE; e += blockDim.x) {
    const int c = e / NPIX, p = e - c * NPIX;
    out[e] = norm_px(img[p * 3 + c]);
  }
}

// one ImagingHorizontalBoxBlur pass along x (vertical == 0) or along y (vertical == 1): src -> dst (both HWC uint8 in LDS)
__device__ void box_pass(const unsigned char* src, unsigned char* dst, int radius, unsigned ww, unsigned fw, int vertical) {
  for (int e = threadIdx.x; e < NBYTE; e += blockDim.x) {
    const int p = e / 3, c = e - p * 3, y = p / S, x = p - y * S;
    const int pos = vertical ? y : x;
    const int stride = vertical ? S * 3 : 3;
    const unsigned char* line = src + (vertical ? x * 3 : y * S * 3) + c;
    unsigned acc = 0;
    for (int k = -radius; k <= radius; ++k) acc += line[min(max(pos + k, 0), S - 1) * stride];
    const unsigned far = line[min(max(pos - radius - 1, 0), S - 1) * stride] + line[min(max(pos + radius + 1, 0), S - 1) * stride];
    const unsigned long long bulk = (unsigned long long)acc * ww + (unsigned long long)far * fw;
    dst[e] = (unsigned char)((bulk + (1u << 23)) >> 24);
  }
}

__global__ __launch_bounds__(1024) void augment_kernel(const unsigned char* __restrict__ images, const int* __restrict__ params,
                                                      const int* __restrict__ table, int B, int K, float* __restrict__ views) {
  extern __shared__ unsigned char lds[];
  unsigned char* bufA = lds;
  unsigned char* bufB = lds + NBYTE;
  unsigned char* tmp = lds + 2 * NBYTE;
  __shared__ int red[16];
  __shared__ int pr[PW];
  const int b = blockIdx.x / K, k = blockIdx.x % K;
  if (threadIdx.x < PW) pr[threadIdx.x] = params[((size_t)b * K + k) * PW + threadIdx.x];
  __syncthreads();
  const int ci = pr[0], cj = pr[1], ch = pr[2], cw = pr[3], flags = pr[4], order = pr[5];
  const unsigned char* src = images + (size_t)b * 3 * NPIX;

  // ---- crop + horizontal pass: tmp[y][xx][c], y < ch (Resample.c ImagingResampleHorizontal_8bpc) ----
  const int* tw = table + (size_t)cw * S * TW;
  for (int e = threadIdx.x; e < ch * S * 3; e += blockDim.x) {
    const int c = e % 3, q = e / 3, xx = q % S, y = q / S;
    const unsigned char* row = src + ((size_t)c * S + (ci + y)) * S + cj;
    int v;
    if (cw == S) {
      v = row[xx];
    } else {
      const int* t = tw + xx * TW;
      int acc = 1 << (PREC - 1);
      for (int n = 0; n < t[1]; ++n) acc += (int)row[t[0] + n] * t[2 + n];
      v = clip8(acc >> PREC);
    }
    tmp[(y * S + xx) * 3 + c] = (unsigned char)v;
  }
  __syncthreads();
  // ---- vertical pass (+ horizontal flip of the result) -> bufA ----
  const int* th = table + (size_t)ch * S * TW;
  for (int e = threadIdx.x; e < NBYTE; e += blockDim.x) {
    const int c = e % 3, q = e / 3, xx = q % S, yy = q / S;
    int v;
    if (ch == S) {
      v = tmp[(yy * S + xx) * 3 + c];
    } else {
      const int* t = th + yy * TW;
      int acc = 1 << (PREC - 1);
      for (int n = 0; n < t[1]; ++n) acc += (int)tmp[((t[0] + n) * S + xx) * 3 + c] * t[2 + n];
      v = clip8(acc >> PREC);
    }
    const int xo = (flags & 1) ? (S - 1 - xx) : xx;
    bufA[(yy * S + xo) * 3 + c] = (unsigned char)v;
  }
  __syncthreads();
  write_view(bufA, views + ((size_t)(2 * k) * B + b) * NBYTE);
  __syncthreads();                                           // the colour operations below rewrite bufA in place

  // ---- ColorJitter in its sampled order ----
  if (flags & 2) {
    for (int step = 0; step < 4; ++step) {
      const int fn = (order >> (2 * step)) & 3;
      if (fn == 3) {                                         // hue: HSV round trip with the uint8 hue channel shifted (wrap-around)
        const int shift = pr[9];
        for (int p = threadIdx.x; p < NPIX; p += blockDim.x) {
          int h, s, v, r, g, bl;
          rgb2hsv(bufA[p * 3], bufA[p * 3 + 1], bufA[p * 3 + 2], h, s, v);
          hsv2rgb((h + shift) & 255, s, v, r, g, bl);
          bufA[p * 3] = (unsigned char)r; bufA[p * 3 + 1] = (unsigned char)g; bufA[p * 3 + 2] = (unsigned char)bl;
        }
      } else {
        const float a = __int_as_float(pr[6 + fn]);
        const bool interp = (a >= 0.0f && a <= 1.0f);
        int mean = 0;
        if (fn == 1) {                                       // contrast: degenerate = int(mean luma + 0.5)
          int part = 0;
          for (int p = threadIdx.x; p < NPIX; p += blockDim.x) part += luma(bufA[p * 3], bufA[p * 3 + 1], bufA[p * 3 + 2]);
          part = (int)wave_sum((float)part);                 // exact: partial sums < 2^24
          if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
          __syncthreads();
          int tot = 0;
          for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) tot += red[wv];
          mean = (int)((double)tot / (double)NPIX + 0.5);
          __syncthreads();
        }
        for (int p = threadIdx.x; p < NPIX; p += blockDim.x) {
          const int r = bufA[p * 3], g = bufA[p * 3 + 1], bl = bufA[p * 3 + 2];
          int dr, dg, db;
          if (fn == 0) dr = dg = db = 0;                     // brightness: black
          else if (fn == 1) dr = dg = db = mean;
          else dr = dg = db = luma(r, g, bl);                // saturation: the pixel's own luma
          bufA[p * 3] = (unsigned char)blend(dr, r, a, interp);
          bufA[p * 3 + 1] = (unsigned char)blend(dg, g, a, interp);
          bufA[p * 3 + 2] = (unsigned char)blend(db, bl, a, interp);
        }
      }
      __syncthreads();
    }
  }
  if (flags & 4) {                                           // RandomGrayscale: convert('L') replicated
    for (int p = threadIdx.x; p < NPIX; p += blockDim.x) {
      const int l = luma(bufA[p * 3], bufA[p * 3 + 1], bufA[p * 3 + 2]);
      bufA[p * 3] = bufA[p * 3 + 1] = bufA[p * 3 + 2] = (unsigned char)l;
    }
    __syncthreads();
  }
  if (pr[13]) {                                              // GaussianBlur: 3 box passes along x, then 3 along y
    const int radius = pr[10];
    const unsigned ww = (unsigned)pr[11], fw = (unsigned)pr[12];
    unsigned char* a = bufA; unsigned char* o = bufB;
    for (int pass = 0; pass < 6; ++pass) {
      box_pass(a, o, radius, ww, fw, pass >= 3);
      __syncthreads();
      unsigned char* t = a; a = o; o = t;
    }                                                        // 6 swaps: the result is back in bufA
  }
  if (flags & 8) {                                           // Solarization(threshold 128)
    for (int e = threadIdx.x; e < NBYTE; e += blockDim.x) { const int v = bufA[e]; bufA[e] = (unsigned char)(v < 128 ? v : 255 - v); }
    __syncthreads();
  }
  write_view(bufA, views + ((size_t)(2 * k + 1) * B + b) * NBYTE);
}

}  // namespace

extern "C" int lafs_augment_views(const uint8_t* images, const int32_t* params, const int32_t* table, int B, int K, float* views,
                                  hipStream_t stream) {
  LAFS_CLEAR_ERROR();
  LAFS_CHECK_ARG(images && params && table && views && B > 0 && K > 0, "bad operand");
  const size_t lds = 3 * (size_t)NBYTE;
  {                                                   // per call (per-device attribute, cheap call): no process-wide flag
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(augment_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { lafs_set_error("lafs_augment_views: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e)); return (int)e; }
  }
  // one workgroup per CU (3 x 37 KB of LDS): 16 waves each to keep the CU busy
  hipLaunchKernelGGL(augment_kernel, dim3(B * K), dim3(1024), lds, stream, images, params, table, B, K, views);
  LAFS_LAUNCH_CHECK();
  return LAFS_OK;
}
