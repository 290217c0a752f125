// Ping-pong between TWO workgroups of one launch: one-way latency of "32 KB payload + flag" hand-offs on gfx950, for a pair on the SAME XCD
// and a pair on DIFFERENT XCDs, with the cache-bypass modifiers each case needs (MI355X_MICROARCH.md: per-XCD L2s are not coherent with each
// other; within an XCD the L2 is the coherence point of its CUs, only the CU's L1 has to be bypassed).
//   far   payload global_store_dwordx4 sc0 sc1 -> s_waitcnt vmcnt(0) -> barrier -> flag store sc0 sc1; poll + loads sc0 sc1      (any pair)
//   near  payload global_store_dwordx4 sc0     -> s_waitcnt vmcnt(0) -> barrier -> flag store sc0    ; poll + loads sc0          (same XCD only)
// Every word is checked every round (payload changes every round).  The XCD of a workgroup is read from HW_REG_XCC_ID.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));

template <int NEAR> __device__ __forceinline__ void st16(void* p, v4i v) {
    if (NEAR == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0\n\ts_nop 3" ::"v"(p), "v"(v) : "memory");
    else if (NEAR == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 3" ::"v"(p), "v"(v) : "memory");
    else if (NEAR == 3) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 3" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 3" ::"v"(p), "v"(v) : "memory");
}
template <int NEAR> __device__ __forceinline__ void ld16x8(const char* p, v4i (&v)[8]) {
    if (NEAR == 1)
        asm volatile(
            "global_load_dwordx4 %0, %8, off sc0\n\tglobal_load_dwordx4 %1, %9, off sc0\n\t"
            "global_load_dwordx4 %2, %10, off sc0\n\tglobal_load_dwordx4 %3, %11, off sc0\n\t"
            "global_load_dwordx4 %4, %12, off sc0\n\tglobal_load_dwordx4 %5, %13, off sc0\n\t"
            "global_load_dwordx4 %6, %14, off sc0\n\tglobal_load_dwordx4 %7, %15, off sc0\n\ts_waitcnt vmcnt(0)"
            : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
            : "v"(p), "v"(p + 4096), "v"(p + 8192), "v"(p + 12288), "v"(p + 16384), "v"(p + 20480), "v"(p + 24576), "v"(p + 28672) : "memory");
    else if (NEAR == 2 || NEAR == 3)
        asm volatile(
            "global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\t"
            "global_load_dwordx4 %2, %10, off sc1\n\tglobal_load_dwordx4 %3, %11, off sc1\n\t"
            "global_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\t"
            "global_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\ts_waitcnt vmcnt(0)"
            : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
            : "v"(p), "v"(p + 4096), "v"(p + 8192), "v"(p + 12288), "v"(p + 16384), "v"(p + 20480), "v"(p + 24576), "v"(p + 28672) : "memory");
    else
        asm volatile(
            "global_load_dwordx4 %0, %8, off sc0 sc1\n\tglobal_load_dwordx4 %1, %9, off sc0 sc1\n\t"
            "global_load_dwordx4 %2, %10, off sc0 sc1\n\tglobal_load_dwordx4 %3, %11, off sc0 sc1\n\t"
            "global_load_dwordx4 %4, %12, off sc0 sc1\n\tglobal_load_dwordx4 %5, %13, off sc0 sc1\n\t"
            "global_load_dwordx4 %6, %14, off sc0 sc1\n\tglobal_load_dwordx4 %7, %15, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
            : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
            : "v"(p), "v"(p + 4096), "v"(p + 8192), "v"(p + 12288), "v"(p + 16384), "v"(p + 20480), "v"(p + 24576), "v"(p + 28672) : "memory");
}
template <int NEAR> __device__ __forceinline__ void stFlag(int* f, int v) {
    if (NEAR == 1) asm volatile("global_store_dword %0, %1, off sc0" ::"v"(f), "v"(v) : "memory");
    else if (NEAR == 2 || NEAR == 3) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(f), "v"(v) : "memory");
    else asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(f), "v"(v) : "memory");
}
template <int NEAR> __device__ __forceinline__ int ldFlag(const int* f) {
    int v;
    if (NEAR == 1) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(f) : "memory");
    else if (NEAR == 2 || NEAR == 3) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(f) : "memory");
    else asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(f) : "memory");
    return v;
}
__device__ __forceinline__ int xccId() {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15;
}
__device__ __forceinline__ double pattern(int it, int i) { return (double)(it * 1315423911u % 1000003u) + 1e-3 * i; }

struct Args { double* payload; int* flags; int* xcc; int* mismatches; long long* cycles; int peer, iters; };

// workgroups 0 and `peer` play; payload[0] goes 0 -> peer in odd half-rounds, payload[1] peer -> 0; flags[0], flags[1] count half-rounds
template <int NEAR>
__global__ __launch_bounds__(256) void k_pingpong(Args a) {
    const int tid = threadIdx.x;
    if (tid == 0) a.xcc[blockIdx.x] = xccId();
    const bool first = blockIdx.x == 0, second = (int)blockIdx.x == a.peer;
    if (!first && !second) return;
    double* out = a.payload + (first ? 0 : 4096);
    const double* in = a.payload + (first ? 4096 : 0);
    int* fo = a.flags + (first ? 0 : 1);
    const int* fi = a.flags + (first ? 1 : 0);
    long long t0 = 0;
    int bad = 0;
    for (int it = 1; it <= a.iters; ++it) {
        if (it == 21 && first && tid == 0) t0 = wall_clock64();
        if (second || it > 1) {
            // wait for the other side's round, read and check its payload
            if (tid == 0) {
                const int want = first ? it - 1 : it;
                long long w0 = wall_clock64();
                while (ldFlag<NEAR>(fi) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    if (wall_clock64() - w0 > 200000LL) break;  // 2 ms
                }
            }
            __syncthreads();
            v4i v[8];
            ld16x8<NEAR>(reinterpret_cast<const char*>(in) + 16 * tid, v);
            const int rit = first ? it - 1 : it;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = 2 * tid + 512 * u;
                const double v0 = __hiloint2double(v[u].y, v[u].x), v1 = __hiloint2double(v[u].w, v[u].z);
                bad += (v0 != pattern(rit * 2 + (first ? 1 : 0), i)) + (v1 != pattern(rit * 2 + (first ? 1 : 0), i + 1));
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = 2 * tid + 512 * u;
            const double v0 = pattern(it * 2 + (first ? 0 : 1), i), v1 = pattern(it * 2 + (first ? 0 : 1), i + 1);
            v4i v;
            v.x = __double2loint(v0); v.y = __double2hiint(v0); v.z = __double2loint(v1); v.w = __double2hiint(v1);
            st16<NEAR>(out + i, v);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) stFlag<NEAR>(fo, it);
    }
    if (first && tid == 0) a.cycles[0] = wall_clock64() - t0;
    if (bad) atomicAdd(a.mismatches, bad);
}

template <int NEAR>
int run(int peer, const char* what) {
    Args a{};
    const int iters = 220, G = 16;
    CK(hipMalloc(&a.payload, 8 * 8192)); CK(hipMalloc(&a.flags, 64)); CK(hipMalloc(&a.xcc, 4 * G)); CK(hipMalloc(&a.mismatches, 4));
    CK(hipMalloc(&a.cycles, 8));
    CK(hipMemset(a.flags, 0, 64)); CK(hipMemset(a.mismatches, 0, 4)); CK(hipMemset(a.payload, 0, 8 * 8192));
    a.peer = peer; a.iters = iters;
    hipLaunchKernelGGL(k_pingpong<NEAR>, dim3(G), dim3(256), 0, 0, a);
    CK(hipDeviceSynchronize());
    int xcc[16], mm; long long cyc;
    CK(hipMemcpy(xcc, a.xcc, 4 * G, hipMemcpyDeviceToHost)); CK(hipMemcpy(&mm, a.mismatches, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&cyc, a.cycles, 8, hipMemcpyDeviceToHost));
    printf("%-52s workgroups 0 (XCC %d) <-> %2d (XCC %d): one-way 32 KB + flag %.2f us; mismatching words %d\n", what, xcc[0], peer, xcc[peer],
           cyc / 100.0 / (2.0 * (iters - 20)), mm);
    fflush(stdout);
    hipFree(a.payload); hipFree(a.flags); hipFree(a.xcc); hipFree(a.mismatches); hipFree(a.cycles);
    return 0;
}
int main() {
    run<0>(1, "system scope (sc0 sc1), neighbours in the grid");
    run<0>(8, "system scope (sc0 sc1), workgroups 0 and 8");
    run<2>(8, "agent scope (sc1 on everything), workgroups 0 and 8");
    run<2>(1, "agent scope (sc1 on everything), neighbours");
    run<3>(8, "plain payload stores + sc1 flag / loads, 0 and 8");
    run<3>(1, "plain payload stores + sc1 flag / loads, neighbours");
    run<1>(8, "workgroup scope (sc0 only), 0 and 8 (expect stale data)");
    return 0;
}
