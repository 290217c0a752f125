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
__device__ __forceinline__ v4i ld16(const void* p) {
    v4i v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
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
                a.tPub[it] = wall_clock64();
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
                int n = 0;
                for (int i = 2 * tid; i < a.words && n < 8; i += 512) v[n++] = ld16(buf + i);
                drain();
                n = 0;
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
                atomicMax((unsigned long long*)&a.lat[it], (unsigned long long)(t - a.tPub[it]));
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
            hipLaunchKernelGGL(k_handoff, dim3(G), dim3(256), 0, 0, a);
            CK(hipDeviceSynchronize());
            std::vector<long long> lat(iters);
            int mm = 0;
            CK(hipMemcpy(lat.data(), a.lat, 8 * iters, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&mm, a.mismatches, 4, hipMemcpyDeviceToHost));
            double sum = 0, mx = 0;
            for (int i = 20; i < iters; ++i) { sum += lat[i]; mx = lat[i] > mx ? lat[i] : mx; }
            printf("%2d KB payload, %3d consumers, %-34s publish -> last consumer done: mean %.2f us, max %.2f us; mismatching words: %d\n", kb,
                   G - 1, names[proto], sum / (iters - 20) / 100.0, mx / 100.0, mm);
            hipFree(a.payload); hipFree(a.flag); hipFree(a.acks); hipFree(a.mismatches); hipFree(a.tPub); hipFree(a.lat); hipFree(a.tDone);
        }
    }
    return 0;
}
