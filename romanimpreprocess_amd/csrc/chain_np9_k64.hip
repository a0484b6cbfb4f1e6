// Wave-specialised fused kernel for f64 ipc4d coefficients, 9 Legendre planes (chain2_kernel.h, KT = double): the narrow form
// (128-column workgroups without the K ring, three per CU); -DC2_K64_NARROW=0 builds the 256-column form for A/B runs.
#include "chain2_kernel.h"

#ifndef C2_K64_NARROW
#define C2_K64_NARROW 1
#endif

// returns the launch status, or 1 when no specialised instantiation fits (the caller takes the stage kernels)
int rip_launch_chain2_k64_np9(rip_ctx *ctx, const RipPlan *plan, const ChainArgs &a) {
    if (a.ngrp == 8) return launch_chain2<9, 8, double, C2_K64_NARROW>(ctx, plan, a);
    if (a.ngrp == 6) return launch_chain2<9, 6, double, C2_K64_NARROW>(ctx, plan, a);
    if (a.ngrp == 16) return launch_chain2<9, 16, double, 2>(ctx, plan, a);   // (two 128-column workgroups per CU: 76 KB each)
    return 1;
}
