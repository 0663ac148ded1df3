// Internal interface between gemm.hip (lafs_gemm_nt dispatch) and gemm_ntw.hip (the wide-tile, one-wave-per-SIMD NT kernel).
#pragma once
#include "lafs_hip.h"

// true when the request is one the wide-tile kernel takes over: K >= 512 (multiple of 32), N >= 576 (multiple of 8), plain /
// GELU / GELU' / residual epilogue (dropout and DropPath scale included), no K split, and at least two rounds of 192 x 192 tiles
// that use >= 85 % of the CU slots (LAFS_NTW=0 switches it off for A/B runs; =2 takes every shape it can compute: tests)
bool lafs_ntw_eligible(const lafs_gemm_nt_args* g);
int lafs_ntw_launch(const lafs_gemm_nt_args* g, hipStream_t stream);
