// Instantiations of the wave-specialised fused kernel for 9 Legendre planes (chain2_kernel.h).
#include "chain2_kernel.h"

#ifndef C2_G16_NARROW   // 16 groups: 2 = 128-column workgroups without K / word rings (three per CU), 0 = the 256-column form
#define C2_G16_NARROW 2
#endif

int rip_launch_chain2_k64_np9(rip_ctx *ctx, const RipPlan *plan, const ChainArgs &a);  // chain_np9_k64.hip

// returns the launch status, or 1 when no specialised instantiation fits (the caller takes the stage kernels)
int rip_launch_chain_np9(rip_ctx *ctx, const RipPlan *plan, const ChainArgs &a, int k_dtype) {
    // wave-specialised kernel: f32 ipc4d with 6, 8 or 16 groups; f64 ipc4d with 6 or 8 groups
    // (merged_dq < 0: this CALDIR set's flag words cannot be merged, RipCal)
    if (!ctx->use_chain2 || a.merged_dq < 0) return 1;
    int rc = 1;
    // (8 / 6 groups with f32 ipc4d: the 256-column form; the narrow forms measured slower there, profiles/r03_summary.md)
    if (k_dtype == RIP_F64) {
        rc = rip_launch_chain2_k64_np9(ctx, plan, a);
    } else {
        if (a.ngrp == 8) rc = launch_chain2<9, 8>(ctx, plan, a);
        if (a.ngrp == 6) rc = launch_chain2<9, 6>(ctx, plan, a);
        if (a.ngrp == 16) rc = launch_chain2<9, 16, float, C2_G16_NARROW>(ctx, plan, a);
    }
    if (rc != 1) ctx->last_form = 2;
    return rc;
}
