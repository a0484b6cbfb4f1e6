// Dispatch of the fused chain kernel (chain2_kernel.h; instantiated per Legendre order in chain_np*.hip).
#include "rip_common.h"

int rip_launch_chain_np4(rip_ctx *ctx, const RipPlan *plan, const ChainArgs &a, int k_dtype);
int rip_launch_chain_np9(rip_ctx *ctx, const RipPlan *plan, const ChainArgs &a, int k_dtype);
int rip_launch_chain_np11(rip_ctx *ctx, const RipPlan *plan, const ChainArgs &a, int k_dtype);

// f32 gain; 4 / 9 / 11 Legendre planes (P_ORDER 3 / 8 / 10); 6, 8 or 16 groups (everything else: the stage kernels)
bool rip_chain_supported(const rip_ctx *ctx, int nplanes, int G, int k_dtype, int gain_dtype) {
    if (gain_dtype != RIP_F32) return false;
    if (nplanes != 4 && nplanes != 9 && nplanes != 11) return false;
    if (G != 6 && G != 8 && G != 16) return false;
    (void)ctx;
    return true;
}

// returns the launch status, or 1 when no fused kernel fits this plan / CALDIR set (the caller then takes the stage kernels)
int rip_launch_chain(rip_ctx *ctx, const RipPlan *plan, const ChainArgs &a, int nplanes, int k_dtype) {
    switch (nplanes) {
        case 4:
            return rip_launch_chain_np4(ctx, plan, a, k_dtype);
        case 9:
            return rip_launch_chain_np9(ctx, plan, a, k_dtype);
        case 11:
            return rip_launch_chain_np11(ctx, plan, a, k_dtype);
    }
    return 1;
}
