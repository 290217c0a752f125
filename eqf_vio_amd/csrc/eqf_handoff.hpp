// Inter-workgroup hand-off INSIDE one launch (gfx950): a producer workgroup publishes a record and a flag, consumer
// workgroups of the same launch wait for the flag and read the record.
//
// Protocol (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility", valid form
// "{sc0 sc1 stores and loads both sides}" with a drained flag): the per-XCD L2s are not coherent and a CU's L1 is never
// refreshed by another CU's stores, so
//   producer:  payload as 16-byte global_store_dwordx4 sc0 sc1 (write-through) -> s_waitcnt vmcnt(0) by every storing
//              thread -> workgroup barrier -> ONE relaxed agent-scope atomic store of the flag;
//   consumer:  one lane polls the flag with relaxed agent-scope loads (s_sleep between polls, bounded by a timeout) ->
//              workgroup barrier -> payload as 16-byte global_load_dwordx4 sc0 sc1 (never served from L1 / a stale line).
// Measured on MI355X with scripts/micro/handoff.hip (one producer, 163 consumer workgroups, every word checked over 200
// rounds): 32 KB from flag store to the LAST consumer holding the whole payload 2.6 us, no stale word; the same with 8-byte
// agent atomics 6.6 us, with plain stores + agent release / acquire fences 7.5 us.
// Flags carry an EPOCH (the update counter of the handle): nothing has to be reset between updates.
#pragma once
#include "eqf_device.hpp"

namespace eqf {

typedef int v4i32 __attribute__((ext_vector_type(4)));

#define EQF_DEV __device__ __forceinline__
EQF_DEV void hoStore16(void* p, double a, double b) {
    v4i32 v;
    v.x = __double2loint(a); v.y = __double2hiint(a); v.z = __double2loint(b); v.w = __double2hiint(b);
    // The trailing s_nop covers the "VMEM store of more than 64 bits -> VALU write of its data VGPRs" hazard: the store reads
    // its data registers a few cycles after issue, and the compiler's hazard recogniser does not look inside inline asm.
    // (Seen without it: the low dword of some stored doubles replaced by whatever the next instruction put in the register --
    // relative errors of 2^-20 in a handful of entries.)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 3" ::"v"(p), "v"(v) : "memory");
}
EQF_DEV void hoDrain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// one thread, after the barrier that follows every storing thread's hoDrain()
EQF_DEV void hoPublish(int* flag, int epoch) { __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// one lane; false = timed out (the producer never came: the caller raises the error flag instead of hanging the GPU).
// `err` (the handle's sticky device error word): a timeout sets bit 128 (kHoErrTimeout: a bit of its own -- bit 8 is a NUMERIC condition of one
// filter, antipodal vectors in the innovation lift, and must not switch the other filters of a batch handle off) there AT ONCE, and every wait looks at that bit every few microseconds
// and gives up as soon as it is set -- so ONE timeout anywhere in a launch unwinds the whole launch in microseconds instead of every
// dependent wait rediscovering it after its own 0.5 s (a launch of 70 000 workgroups would spin for hours; the advisor's round-3 finding).
constexpr int kHoErrTimeout = 128;
#ifdef EQF_WAIT_STATS
// instrumented build (scripts/wait_stats.py): per role class of k_chol_resident -- 0 H, 1 T, 2 W, 3 F0, 4 prep, 5 downdate tile; 8 + class for the
// E-chain -- the 100 MHz ticks its workgroups spent inside hoWait / the downdate's gate, their lifetimes, and how many there were
__device__ unsigned long long g_waitStats[16][3];
__shared__ int sStatClass;
__shared__ unsigned long long sStatWait;
#define EQF_STAT_CLASS(c) do { if (threadIdx.x == 0) sStatClass = (c); } while (0)
#define EQF_STAT_WAIT(t) atomicAdd(&sStatWait, (unsigned long long)(t))
#else
#define EQF_STAT_CLASS(c) do { } while (0)
#define EQF_STAT_WAIT(t) do { } while (0)
#endif
EQF_DEV bool hoAborted(const int* err) { return err && (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kHoErrTimeout); }
EQF_DEV bool hoWait(const int* flag, int epoch, int* err = nullptr) {
    const long long t0 = wall_clock64();  // 100 MHz
    int polls = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
#ifndef EQF_POLL_SLEEP
#define EQF_POLL_SLEEP 1
#endif
        __builtin_amdgcn_s_sleep(EQF_POLL_SLEEP);
        if ((++polls & 255) == 0) {
            if (hoAborted(err)) return false;
            if (wall_clock64() - t0 > 50000000LL) {  // 0.5 s (a launch for N = 4000 runs 60 ms)
                if (err) atomicOr(err, kHoErrTimeout);
                return false;
            }
        }
    }
    EQF_STAT_WAIT(wall_clock64() - t0);
    return true;
}
// Ten 16-byte sc0 sc1 loads at p + 4096 k (k = 0..9: 40 KB per 256-thread workgroup, one diagonal-factor record), all in
// flight together and COMPLETE on return: the s_waitcnt sits inside the same asm statement because the compiler does not
// track the asynchronous register write of an inline-asm load and could otherwise copy a result register too early.
EQF_DEV void hoLoad16x10(const char* p, v4i32 (&v)[10]) {
    const char *p0 = p, *p1 = p + 4096, *p2 = p + 2 * 4096, *p3 = p + 3 * 4096, *p4 = p + 4 * 4096, *p5 = p + 5 * 4096,
               *p6 = p + 6 * 4096, *p7 = p + 7 * 4096, *p8 = p + 8 * 4096, *p9 = p + 9 * 4096;
    asm volatile(
        "global_load_dwordx4 %0, %10, off sc0 sc1\n\tglobal_load_dwordx4 %1, %11, off sc0 sc1\n\t"
        "global_load_dwordx4 %2, %12, off sc0 sc1\n\tglobal_load_dwordx4 %3, %13, off sc0 sc1\n\t"
        "global_load_dwordx4 %4, %14, off sc0 sc1\n\tglobal_load_dwordx4 %5, %15, off sc0 sc1\n\t"
        "global_load_dwordx4 %6, %16, off sc0 sc1\n\tglobal_load_dwordx4 %7, %17, off sc0 sc1\n\t"
        "global_load_dwordx4 %8, %18, off sc0 sc1\n\tglobal_load_dwordx4 %9, %19, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]), "=&v"(v[9])
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7), "v"(p8), "v"(p9)
        : "memory");
}
EQF_DEV double hoLoad8(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
EQF_DEV void hoStore8(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
EQF_DEV double hoLo(const v4i32& v) { return __hiloint2double(v.y, v.x); }
EQF_DEV double hoHi(const v4i32& v) { return __hiloint2double(v.w, v.z); }

}  // namespace eqf
