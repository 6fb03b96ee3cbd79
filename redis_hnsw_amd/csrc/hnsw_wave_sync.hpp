// hnsw_wave_sync.hpp -- the synchronisation points of code that ONE wavefront runs on its own.
//
// The insert / delete kernels (hnsw_insert.hpp, hnsw_occ.hpp, hnsw_occ_par.hpp, hnsw_plan_lean.hpp) are written for one
// wavefront per copy of the code: one-wave workgroups, or several wavefronts that each run the code on their own and meet
// only at explicit hand-overs (team_bar, the two-wave plans' mailbox).  Where the lanes of that wave hand each other
// data they call wave_sync().  The data travels through LDS, and through HBM as well -- words of adjacency rows, the
// journal, plan and read-log entries -- so the synchronisation has to wait for the wave's own STORES TO MEMORY too:
//
//     s_waitcnt vmcnt(0) lgkmcnt(0)  +  wave barrier  +  compiler memory barrier.
//
// A workgroup-scope fence does not do that on this target (it waits for LDS only), and a __syncthreads() of a 64-thread
// workgroup is dropped by the compiler together with the s_waitcnt vmcnt(0) it otherwise puts in front of an s_barrier:
// round 4's dim-768 commit accepted stale re-selections because its validation read journal entries another lane's store
// had not landed yet (scripts/del_repro.py).  Until round 5 this header redefined __syncthreads() for the units that
// included it first; now the shared code names what it means, and tests/test_capi_cpu.py checks in the ISA of an insert
// unit that every wave barrier is preceded by the full wait.
//
// The code of hnsw_device.hpp is shared with the SEARCH kernels (and the fast build's), whose lanes hand over through LDS
// only: there the synchronisation point is dev_sync(), and every translation unit says which of the two it is built for
// (HNSW_SYNC_WAVE_FULL / HNSW_SYNC_BLOCK, see hnsw_device.hpp) -- a unit that says nothing does not compile.
#pragma once
#include <hip/hip_runtime.h>
namespace hnsw {
__device__ __forceinline__ void wave_sync()
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void wave_sync_full() { wave_sync(); }
// An ordering point for data that stays in LDS (or in registers): one wavefront's LDS accesses are served in issue
// order, so all this does is keep the compiler from moving them across.  Named -- and marked in the ISA -- so that a wave
// barrier without the full wait is a stated decision: tests/test_capi_cpu.py accepts no other kind in an insert unit.
__device__ __forceinline__ void lds_order()
{
#ifdef HNSW_SYNC_WAVE_FULL
    asm volatile("; hnsw lds_order");                     // (a comment in the ISA: no instruction, no constraint on the compiler)
#endif
    __builtin_amdgcn_wave_barrier();
}
} // namespace hnsw
