// Instantiations of the fused chain kernel for 4 Legendre planes (see chain_kernel.h).
#include "chain_kernel.h"

int rip_launch_chain_np4(rip_ctx *ctx, const RipPlan *plan, const ChainArgs &a, int k_dtype) {
    return launch_chain_np<4>(ctx, plan, a, k_dtype);
}
