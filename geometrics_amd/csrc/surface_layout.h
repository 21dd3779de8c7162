// Layout of the surface-loss backward scratch (`order`, int32 words) shared by the fused scan (which writes the
// gradient records) and the finalize / gather passes (csrc/surface_gather.hip):
//   off[b,nf+1] | seg[b,cap] | pface[b,cap] | slot[b,cap] | pad to 4 words | rec[b,cap,2] float4     with cap = num + n_gt
#pragma once
#include <stdint.h>

static inline int64_t geom_surface_order_ints(int b, int nf, int64_t cap)
{
    return (((int64_t)b * (nf + 1) + 3 * (int64_t)b * cap) + 3) & ~3ll;
}
