// Microbenchmark + correctness check of the inter-workgroup hand-off used by the resident update chain (gfx950).
// One producer workgroup publishes a payload of `bytes` (a "diagonal-factor record" / a solved 64x64 panel block) and a flag;
// every other workgroup polls the flag, reads the whole payload, CHECKS EVERY WORD, acknowledges; repeated `iters` times
// with the payload changing every round (so a stale line shows up as a mismatch).  Protocols (MI355X_MICROARCH.md,
// "Workgroup dispatch, XCD placement & inter-workgroup visibility"):
//   A  payload: 8-byte relaxed agent-scope atomic stores (global_store_dwordx2 sc0 sc1) -> s_waitcnt vmcnt(0) -> barrier ->
//      flag: relaxed agent-scope atomic store;   consumer: relaxed agent poll -> barrier -> 8-byte relaxed agent atomic loads
//   B  payload: plain stores -> barrier -> lane-0 release fence (agent) -> s_waitcnt vmcnt(0) -> relaxed flag;
//      consumer: relaxed poll -> agent acquire fence -> barrier -> plain loads
//   C  like A with 16-byte accesses (inline asm global_store_dwordx4 / global_load_dwordx4 ... sc0 sc1)
// Reported: time from the producer's flag store to the last consumer's "payload complete" (100 MHz wall clock), mismatches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void st8(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld8(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st16(void* p, v4i v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
// eight 16-byte sc0 sc1 loads at p + k * stride, all in flight together, returned complete (the s_waitcnt is inside the asm:
// the compiler does not track the asynchronous write of an inline-asm load and may otherwise copy a register too early)
__device__ __forceinline__ void ld16x8(const char* p, long stride, v4i (&v)[8]) {
    const char *p0 = p, *p1 = p + stride, *p2 = p + 2 * stride, *p3 = p + 3 * stride, *p4 = p + 4 * stride, *p5 = p + 5 * stride,
               *p6 = p + 6 * stride, *p7 = p + 7 * stride;
    asm volatile(
        "global_load_dwordx4 %0, %8, off sc0 sc1\n\tglobal_load_dwordx4 %1, %9, off sc0 sc1\n\t"
        "global_load_dwordx4 %2, %10, off sc0 sc1\n\tglobal_load_dwordx4 %3, %11, off sc0 sc1\n\t"
        "global_load_dwordx4 %4, %12, off sc0 sc1\n\tglobal_load_dwordx4 %5, %13, off sc0 sc1\n\t"
        "global_load_dwordx4 %6, %14, off sc0 sc1\n\tglobal_load_dwordx4 %7, %15, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7)
        : "memory");
}
__device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ double pattern(int it, int i) { return (double)(it * 1315423911u % 1000003u) + 1e-3 * i; }

struct Args {
    double* payload;     // [2][words] double-buffered by round parity
    int* flag;           // round counter
    int* acks;           // arrivals of the current round
    long long* tPub;     // producer's publish time per round
    long long* tDone;    // [G] consumer completion time of the last round seen
    long long* lat;      // [iters] max over consumers (atomicMax)
    int* mismatches;
    int words, iters, proto;
};

__global__ __launch_bounds__(256) void k_handoff(Args a) {
    const int tid = threadIdx.x, G = gridDim.x;
    __shared__ int sIt;
    for (int it = 0; it < a.iters; ++it) {
        double* buf = a.payload + (size_t)(it & 1) * a.words;
        if (blockIdx.x == 0) {
            // ---- producer
            if (a.proto == 2) {
                for (int i = 2 * tid; i < a.words; i += 512) {
                    const double v0 = pattern(it, i), v1 = pattern(it, i + 1);
                    v4i v;
                    v.x = __double2loint(v0); v.y = __double2hiint(v0); v.z = __double2loint(v1); v.w = __double2hiint(v1);
                    st16(buf + i, v);
                }
                drain();
            } else if (a.proto == 0) {
                for (int i = tid; i < a.words; i += 256) st8(buf + i, pattern(it, i));
                drain();
            } else {
                for (int i = tid; i < a.words; i += 256) buf[i] = pattern(it, i);
            }
            __syncthreads();
            if (tid == 0) {
                if (a.proto == 1) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    drain();
                }
                __hip_atomic_store(&a.tPub[it], (long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(a.flag, it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // wait for every consumer before the next round (bounded)
                long long t0 = wall_clock64();
                while (__hip_atomic_load(a.acks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (G - 1) * (it + 1)) {
                    __builtin_amdgcn_s_sleep(2);
                    if (wall_clock64() - t0 > 100000000LL) break;  // 1 s
                }
            }
            __syncthreads();
        } else {
            // ---- consumer
            if (tid == 0) {
                long long t0 = wall_clock64();
                while (__hip_atomic_load(a.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < it + 1) {
                    __builtin_amdgcn_s_sleep(1);
                    if (wall_clock64() - t0 > 100000000LL) break;
                }
                if (a.proto == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                sIt = it;
            }
            __syncthreads();
            int bad = 0;
            if (a.proto == 2) {
                v4i v[8];
                // (payloads smaller than 32 KB: the surplus loads re-read the first chunk)
                const long stride = a.words >= 4096 ? 4096 : 0;
                ld16x8(reinterpret_cast<const char*>(buf + ((2 * tid) % a.words)), stride, v);
                int n = 0;
                for (int i = 2 * tid; i < a.words && n < 8; i += 512, ++n) {
                    const double v0 = __hiloint2double(v[n].y, v[n].x), v1 = __hiloint2double(v[n].w, v[n].z);
                    bad += (v0 != pattern(it, i)) + (v1 != pattern(it, i + 1));
                }
            } else {
                double v[16];
                int n = 0;
                for (int i = tid; i < a.words && n < 16; i += 256) v[n++] = a.proto == 0 ? ld8(buf + i) : buf[i];
                n = 0;
                for (int i = tid; i < a.words && n < 16; i += 256, ++n) bad += (v[n] != pattern(it, i));
            }
            if (bad) atomicAdd(a.mismatches, bad);
            __syncthreads();
            if (tid == 0) {
                const long long t = wall_clock64();
                atomicMax((unsigned long long*)&a.lat[it], (unsigned long long)(t - __hip_atomic_load(&a.tPub[it], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
                __hip_atomic_fetch_add(a.acks, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 64, iters = 200;
    const char* names[3] = {"A 8-byte agent atomics", "B plain + release/acquire fences", "C 16-byte sc0 sc1 (asm)"};
    for (int kb : {1, 8, 32}) {
        const int words = kb * 1024 / 8;
        for (int proto = 0; proto < 3; ++proto) {
            Args a{};
            CK(hipMalloc(&a.payload, sizeof(double) * 2 * words));
            CK(hipMalloc(&a.flag, 4)); CK(hipMalloc(&a.acks, 4)); CK(hipMalloc(&a.mismatches, 4));
            CK(hipMalloc(&a.tPub, 8 * iters)); CK(hipMalloc(&a.lat, 8 * iters)); CK(hipMalloc(&a.tDone, 8 * G));
            CK(hipMemset(a.flag, 0, 4)); CK(hipMemset(a.acks, 0, 4)); CK(hipMemset(a.mismatches, 0, 4)); CK(hipMemset(a.lat, 0, 8 * iters));
            a.words = words; a.iters = iters; a.proto = proto;
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_handoff, dim3(G), dim3(256), 0, 0, a);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float kms = 0;
            CK(hipEventElapsedTime(&kms, e0, e1));
            std::vector<long long> lat(iters);
            int mm = 0;
            CK(hipMemcpy(lat.data(), a.lat, 8 * iters, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&mm, a.mismatches, 4, hipMemcpyDeviceToHost));
            double sum = 0, mx = 0;
            for (int i = 20; i < iters; ++i) { sum += lat[i]; mx = lat[i] > mx ? lat[i] : mx; }
            printf("%2d KB payload, %3d consumers, %-34s publish -> last consumer done: mean %.2f us, max %.2f us; mismatching words: %d; whole round (publish + read + ack) %.2f us\n", kb,
                   G - 1, names[proto], sum / (iters - 20) / 100.0, mx / 100.0, mm, kms * 1000.0 / iters);
            hipFree(a.payload); hipFree(a.flag); hipFree(a.acks); hipFree(a.mismatches); hipFree(a.tPub); hipFree(a.lat); hipFree(a.tDone);
        }
    }
    return 0;
}
