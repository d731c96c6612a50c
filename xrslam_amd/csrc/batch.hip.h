// batch.hip.h -- the argument block of a launch that serves several contexts.
//
// The per-frame kernels take their arguments as a block of up to XB argument sets, one per blockIdx.z: a launch serves that many
// contexts (several sequences of an instance group on one GPU, group.hip.h) -- or one, when a context launches alone.  The blocks of
// an entry see exactly the arguments, block indices (x, y) and LDS they would see in a launch of their own, so an entry's results do
// not depend on what it is batched with.  The block travels in the kernel-argument segment (scalar loads; 4 KB at most: the largest
// argument set, k_lk_track's two pyramid views, is 336 bytes).
#pragma once
#include "host_mailbox.hip.h"

namespace xrhip {

constexpr int XB = 8;
template <class A> struct Batch {
    A e[XB];
};

}   // namespace xrhip
