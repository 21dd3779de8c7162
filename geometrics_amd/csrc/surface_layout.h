// Layout of the surface-loss backward scratch (`order`, int32 words) shared by the fused scan (which writes the
// gradient records) and the finalize / gather passes (csrc/surface_gather.hip):
//   off[b,nf+1] | seg[b,cap] | pface[b,cap] | slot[b,cap] | pad to 4 words | rec[b,cap,2] float4 | status[b+1, pad to 4]
// with cap = num + n_gt.  status[i] (i < b): 0 = mesh i's ordering is complete, anything else = the pass that orders it gave up
// (a finalize role of the scan launch whose wait timed out); status[b]: the same for the loss sum.  Written by EVERY finalize
// pass (0 on success), read by the gather backward: a mesh whose ordering is not there gets NaN gradients, never a walk
// through a half-built order.
#pragma once
#include <stdint.h>

static inline int64_t geom_surface_order_ints(int b, int nf, int64_t cap)
{
    return (((int64_t)b * (nf + 1) + 3 * (int64_t)b * cap) + 3) & ~3ll;
}
static inline int64_t geom_surface_status_ints(int b) { return ((int64_t)b + 1 + 3) & ~3ll; }
static inline int64_t geom_surface_status_offset(int b, int nf, int64_t cap) { return geom_surface_order_ints(b, nf, cap) + (int64_t)b * cap * 8; }
