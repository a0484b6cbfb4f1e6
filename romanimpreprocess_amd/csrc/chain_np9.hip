// Instantiations of the fused chain kernel for 9 Legendre planes (see chain_kernel.h).
#include "chain_kernel.h"

int rip_launch_chain_np9(rip_ctx *ctx, const RipPlan *plan, const ChainArgs &a, int k_dtype) {
    return launch_chain_np<9>(ctx, plan, a, k_dtype);
}
