// Instantiations of the fused chain kernels for 9 Legendre planes (chain_kernel.h, chain2_kernel.h).
#include "chain2_kernel.h"

int rip_launch_chain2_k64_np9(rip_ctx *ctx, const RipPlan *plan, const ChainArgs &a);  // chain_np9_k64.hip

int rip_launch_chain_np9(rip_ctx *ctx, const RipPlan *plan, const ChainArgs &a, int k_dtype) {
    // wave-specialised kernel for the common cases (f32 ipc4d: 6, 8 or 16 groups; f64 ipc4d: 6 or 8 groups); general fused
    // kernel otherwise
    if (k_dtype == RIP_F64 && ctx->use_chain2 && a.merged_dq >= 0) {
        const int rc = rip_launch_chain2_k64_np9(ctx, plan, a);
        if (rc != 1) {
            ctx->last_form = 2;
            return rc;
        }
    }
    if (k_dtype == RIP_F32 && ctx->use_chain2 && a.merged_dq >= 0) {   // (merged_dq < 0: this CALDIR set's flag words cannot be merged, RipCal)
        int rc = 1;
        if (a.ngrp == 8) rc = launch_chain2<9, 8>(ctx, plan, a);
        if (a.ngrp == 6) rc = launch_chain2<9, 6>(ctx, plan, a);
        if (a.ngrp == 16) rc = launch_chain2<9, 16>(ctx, plan, a);
        if (rc != 1) {
            ctx->last_form = 2;
            return rc;
        }
    }
    ctx->last_form = 1;
    return launch_chain_np<9>(ctx, plan, a, k_dtype);
}
