// Wave-private fused kernel (chain3_kernel.h), 9 Legendre planes: f64 ipc4d coefficients x 16 groups -- the one configuration
// whose rings do not fit the wave-specialised kernel's workgroup (x ring 48 KB + f64 first-iterate ring 96 KB + f64 K ring 36 KB
// of LDS); every other configuration runs chain2_kernel.h (round 3: the other instantiations of this kernel were dropped).
#include "chain3_kernel.h"

// returns the launch status, or 1 when no instantiation fits (the caller falls back to the other fused kernels)
int rip_launch_chain3_k64_np9(rip_ctx *ctx, const RipPlan *plan, const ChainArgs &a) {
    if (a.ngrp == 16) return launch_chain3<9, 16, double>(ctx, plan, a);
    return 1;
}
