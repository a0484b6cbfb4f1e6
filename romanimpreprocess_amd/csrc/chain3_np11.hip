// Wave-private fused kernel (chain3_kernel.h), f32 ipc4d coefficients, 11 Legendre planes.
#include "chain3_kernel.h"

// returns the launch status, or 1 when no instantiation fits (the caller falls back to the other fused kernels)
int rip_launch_chain3_np11(rip_ctx *ctx, const RipPlan *plan, const ChainArgs &a) {
    if (a.ngrp == 8) return launch_chain3<11, 8>(ctx, plan, a);
    if (a.ngrp == 6) return launch_chain3<11, 6>(ctx, plan, a);
    if (a.ngrp == 16) return launch_chain3<11, 16>(ctx, plan, a);
    return 1;
}
