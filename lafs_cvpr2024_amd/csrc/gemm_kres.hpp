// Internal interface between gemm.hip (lafs_gemm_nt dispatch) and gemm_kres.hip (the K-resident streaming kernel).
#pragma once
#include "lafs_hip.h"

// true when the request is one the K-resident kernel covers: K == 384, N % 64 == 0, N <= 1536, at least 2048 rows, plain /
// GELU / GELU' / residual epilogue, no dropout, no K split (LAFS_KRES=0 in the environment switches it off for A/B runs)
bool lafs_kres_eligible(const lafs_gemm_nt_args* g);
int lafs_kres_launch(const lafs_gemm_nt_args* g, hipStream_t stream);
// The ping-pong variant (gemm_kpp.hip): one 8-wave workgroup per CU, MFMA turn of one half against the epilogue turn of the other.
// lafs_kpp_selected: the request (already K-resident-eligible) takes it (LAFS_KPP bit mask in the environment, as LAFS_KRES).
bool lafs_kpp_selected(const lafs_gemm_nt_args* g);
int lafs_kpp_launch(const lafs_gemm_nt_args* g, hipStream_t stream);
