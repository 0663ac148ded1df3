// Internal interface between gemm.hip (lafs_gemm_nt dispatch) and gemm_big.hip (the 256x256 one-workgroup-per-CU kernel).
#pragma once
#include "lafs_hip.h"

// true when the request is one the 256x256 persistent kernel covers and is expected to win on: bf16 operands, K % 64 == 0,
// K >= 512, wide outputs (N >= 512) on many rows, plain / GELU / GELU' / residual epilogue, no K split
// (LAFS_OPT_NT_BIG = 0 switches it off for A/B runs)
bool lafs_big_eligible(const lafs_gemm_nt_args* g);
int lafs_big_launch(const lafs_gemm_nt_args* g, hipStream_t stream);
