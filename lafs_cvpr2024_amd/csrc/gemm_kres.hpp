// Internal interface between gemm.hip (lafs_gemm_nt dispatch) and gemm_kres.hip (the K-resident streaming kernel).
#pragma once
#include "lafs_hip.h"

// true when the request is one the K-resident kernel covers: K == 384, N % 64 == 0, N <= 1536, at least 2048 rows, plain /
// GELU / GELU' / residual epilogue, no dropout, no K split (LAFS_OPT_KRES_MASK = 0 switches it off for A/B runs)
bool lafs_kres_eligible(const lafs_gemm_nt_args* g);
int lafs_kres_launch(const lafs_gemm_nt_args* g, hipStream_t stream);
