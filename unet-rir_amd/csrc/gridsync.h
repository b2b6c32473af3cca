// gridsync.h - "the last workgroup finishes the reduction": the kernels of a two-stage reduction (per-workgroup partials, then a
// fixed-order sum of the partials) fold the second stage into the first launch.  Every workgroup writes its partials, makes them
// visible device-wide and increments an arrival counter; the workgroup that draws the last ticket re-reads ALL partials and sums
// them in the same fixed order the separate second-stage kernel used - the result does not depend on which workgroup that is,
// so it stays bit-reproducible run to run.  No workgroup ever waits for another one (no spinning, no co-residency requirement).
//
// Visibility on gfx950: the 8 XCDs have separate L2s.  The writers' release fence (agent scope) writes their partials back
// (buffer_wbl2 sc1), the relaxed device-scope atomic is performed at the memory side, and the last workgroup's acquire fence
// (buffer_inv sc1) drops whatever stale lines its own L1 / L2 hold before it reads the partials.
#pragma once
#include <hip/hip_runtime.h>

// Call from ALL threads of the workgroup after its partials have been written.  Returns true (to every thread) in the one workgroup
// that arrives last among `expected`; that workgroup has also reset the counter to zero for the next launch on this stream.
__device__ __forceinline__ bool last_workgroup_arrives(unsigned* __restrict__ counter, unsigned expected) {
    __shared__ unsigned s_last;
    __threadfence();                                   // release, agent scope: this wave's partial stores are written back to memory
    __syncthreads();                                   // ... and so are those of every other wave of the workgroup
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned last = (t == expected - 1u) ? 1u : 0u;
        if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // clean for the next launch
        s_last = last;
    }
    __syncthreads();
    const bool last = s_last != 0u;
    if (last) __threadfence();                         // acquire, agent scope: drop stale lines before reading the others' partials
    return last;
}
