// hnsw_wave_sync.hpp -- what "__syncthreads()" means in the translation units of the insert / delete kernels.
// Include FIRST (before hip_runtime's users) in a unit whose kernels run the shared insert code on ONE wavefront per
// copy: hnsw_tu_insert.hip, hnsw_tu_occ.hip, hnsw_tu_planlean.hip (one-wave workgroups) and hnsw_tu_planduo.hip,
// hnsw_tu_occteam.hip (several wavefronts that each run the code on their own; their only s_barriers are the explicit
// hand-overs).
//
// The shared code writes "__syncthreads()" where the lanes of the wave hand each other data -- through LDS, and through
// HBM as well (words of adjacency rows, the journal, plan and read-log entries).  For a one-wave workgroup the
// compiler drops the s_barrier, and with it the "s_waitcnt vmcnt(0)" it puts in front of every s_barrier on this
// target; what is left are workgroup-scope fences, which wait for LDS only.  A lane's load could then be issued while
// another lane's store to the same words was still in flight.  The four-wave commit on the dim-768 variant is where it
// showed (validation read journal entries before they had landed and accepted stale re-selections:
// scripts/del_repro.py); the one-wave kernels ran the same code with the same weak synchronisation.  Here the
// synchronisation is spelled out: everything outstanding has completed, and the compiler moves no memory access across.
#pragma once
#include <hip/hip_runtime.h>
namespace hnsw {
__device__ __forceinline__ void wave_sync_full()
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}
} // namespace hnsw
#define __syncthreads() ::hnsw::wave_sync_full()
