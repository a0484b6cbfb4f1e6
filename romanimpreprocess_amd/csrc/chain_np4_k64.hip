// Wave-specialised fused kernel for f64 ipc4d coefficients, 4 Legendre planes (chain2_kernel.h, KT = double).
#include "chain2_kernel.h"

// returns the launch status, or 1 when no specialised instantiation fits (the caller falls back to the general fused kernel)
int rip_launch_chain2_k64_np4(rip_ctx *ctx, const RipPlan *plan, const ChainArgs &a) {
    if (a.ngrp == 8) return launch_chain2<4, 8, double>(ctx, plan, a);
    if (a.ngrp == 6) return launch_chain2<4, 6, double>(ctx, plan, a);
    return 1;
}
