// host_mailbox.hip.h -- stores into pinned, device-mapped host memory that the host polls (results + a flag written last).
//
// __threadfence_system() is a sequentially consistent fence at system scope.  On gfx950 it is
//     buffer_wbl2 sc0 sc1 ; s_waitcnt vmcnt(0) lgkmcnt(0) ; buffer_inv sc0 sc1
// -- write back every dirty line of the L2 (whatever the kernels before left there: a response plane, pyramid levels ...), wait, then
// drop the non-local lines of L2 and L1.  A mailbox needs neither the write-back of unrelated lines nor the invalidation: its payload
// is written with system-scope stores (sc0 sc1: through the L2 to the host link), the wavefront waits until they are acknowledged
// (vmcnt counts stores on gfx9), the workgroup meets at a barrier if several wavefronts wrote, and only then is the flag stored.
// (In-kernel timers, profiles/r06_fence.md.)
#pragma once
#include <hip/hip_runtime.h>

namespace xrhip {

__device__ __forceinline__ void host_store(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void host_store(unsigned char *p, unsigned char v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void host_store(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void host_store(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void host_store(double *p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void host_store(float *p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned *>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Used where the payload is host memory only and the fence sat on the kernel's critical path: k_harris_select (fence 2.1 -> 0.45 us
// of a 13 us kernel) and kp_preintegrate (record 2.7 -> 0.7 us).  NOT used where a kernel also leaves results in DEVICE memory that
// another stream's kernel reads once the host has seen the flag (the solves' control blocks and states: kb_chain, kb_solve_try,
// kb_trials_wide keep __threadfence_system()), nor in k_lk_track, where it made no measurable difference (see there).
// this wavefront's host_store()s have been acknowledged
__device__ __forceinline__ void host_stores_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

}   // namespace xrhip
