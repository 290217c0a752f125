// What the two multi-GPU C++ hosts (example_batch_rccl.cpp, example_tiled_rccl.cpp) share: one process per GPU without MPI or Python.
//   * launch: `<exe> --spawn G ...` re-executes itself G times (fork + exec BEFORE anything touches HIP) with EQF_RANK / EQF_WORLD /
//     EQF_ID_FILE in the environment and waits; a launcher of the caller's own (mpirun, srun, torchrun) sets those three instead -- RANK /
//     WORLD_SIZE / LOCAL_RANK are understood too.
//   * rendezvous: rank 0 calls ncclGetUniqueId and publishes the 128 bytes through EQF_ID_FILE (written under a temporary name, then
//     renamed: a reader sees all of it or nothing); the others poll for the file.  One node, one file system: nothing else is needed.
//   * one rank per GPU, checked: RCCL refuses two ranks on one device, so a job with more ranks than visible GPUs stops with a message
//     from every rank before the communicator is built.
// Plumbing only -- the filter is behind include/eqf_vio_amd.h.
#pragma once
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "../../include/eqf_vio_amd.h"
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace eqf_rccl {

#define EQF_HIP(call)                                                                                            \
    do {                                                                                                         \
        hipError_t e_ = (call);                                                                                  \
        if (e_ != hipSuccess) {                                                                                  \
            std::fprintf(stderr, "rank %d: %s failed: %s (%s:%d)\n", eqf_rccl::g_rank, #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            std::exit(4);                                                                                        \
        }                                                                                                        \
    } while (0)
#define EQF_NCCL(call)                                                                                           \
    do {                                                                                                         \
        ncclResult_t r_ = (call);                                                                                \
        if (r_ != ncclSuccess) {                                                                                 \
            std::fprintf(stderr, "rank %d: %s failed: %s (%s:%d)\n", eqf_rccl::g_rank, #call, ncclGetErrorString(r_), __FILE__, __LINE__); \
            std::exit(5);                                                                                        \
        }                                                                                                        \
    } while (0)

inline int g_rank = 0;

inline const char* env_first(std::initializer_list<const char*> names) {
    for (const char* n : names)
        if (const char* v = std::getenv(n)) return v;
    return nullptr;
}

// `--spawn G` as argv[1..2]: start G copies of this executable (same remaining arguments), wait, return the first non-zero exit code.
// Returns -1 when the process is a rank itself and should carry on.
inline int maybe_spawn(int argc, char** argv) {
    if (argc < 3 || std::string(argv[1]) != "--spawn") return -1;
    const int G = std::atoi(argv[2]);
    if (G < 1) {
        std::fprintf(stderr, "--spawn needs a positive rank count\n");
        return 2;
    }
    char idfile[] = "/tmp/eqf_rccl_id_XXXXXX";
    const int fd = mkstemp(idfile);
    if (fd >= 0) close(fd);
    unlink(idfile);  // (rank 0 creates it; the name is what is shared)
    std::vector<pid_t> kids;
    for (int r = 0; r < G; ++r) {
        const pid_t pid = fork();
        if (pid == 0) {
            setenv("EQF_RANK", std::to_string(r).c_str(), 1);
            setenv("EQF_WORLD", std::to_string(G).c_str(), 1);
            setenv("EQF_ID_FILE", idfile, 1);
            setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);  // dmabuf IPC (the only kind the host driver of this pool supports)
            std::vector<char*> av;
            av.push_back(argv[0]);
            for (int i = 3; i < argc; ++i) av.push_back(argv[i]);
            av.push_back(nullptr);
            execv("/proc/self/exe", av.data());
            std::perror("execv");
            _exit(127);
        }
        kids.push_back(pid);
    }
    int rc = 0;
    for (pid_t p : kids) {
        int st = 0;
        waitpid(p, &st, 0);
        const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + WTERMSIG(st);
        if (code != 0 && rc == 0) rc = code;
    }
    unlink(idfile);
    return rc;
}

struct World {
    int rank = 0, world = 1, device = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
};

inline World init_world() {
    World w;
    const char* r = env_first({"EQF_RANK", "RANK", "OMPI_COMM_WORLD_RANK"});
    const char* n = env_first({"EQF_WORLD", "WORLD_SIZE", "OMPI_COMM_WORLD_SIZE"});
    w.rank = r ? std::atoi(r) : 0;
    w.world = n ? std::atoi(n) : 1;
    g_rank = w.rank;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
    if (ndev < w.world) {
        std::fprintf(stderr, "rank %d of %d: not enough devices -- %d ranks need %d visible GPUs (one process per GPU), this process sees %d\n",
            w.rank, w.world, w.world, w.world, ndev);
        std::exit(3);
    }
    const char* lr = env_first({"EQF_LOCAL_RANK", "LOCAL_RANK"});
    w.device = lr ? std::atoi(lr) : w.rank;
    EQF_HIP(hipSetDevice(w.device));
    EQF_HIP(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
    ncclUniqueId id;
    const char* idf = std::getenv("EQF_ID_FILE");
    if (w.world == 1 && !idf) {
        EQF_NCCL(ncclGetUniqueId(&id));
    } else {
        if (!idf) {
            std::fprintf(stderr, "rank %d: EQF_ID_FILE is not set (start the job with --spawn G, or export it in your launcher)\n", w.rank);
            std::exit(2);
        }
        if (w.rank == 0) {
            EQF_NCCL(ncclGetUniqueId(&id));
            const std::string tmp = std::string(idf) + ".tmp";
            FILE* f = std::fopen(tmp.c_str(), "wb");
            if (!f || std::fwrite(&id, sizeof(id), 1, f) != 1) {
                std::fprintf(stderr, "rank 0: cannot write %s\n", tmp.c_str());
                std::exit(2);
            }
            std::fclose(f);
            std::rename(tmp.c_str(), idf);
        } else {
            const auto t0 = std::chrono::steady_clock::now();
            for (;;) {
                FILE* f = std::fopen(idf, "rb");
                if (f) {
                    const size_t got = std::fread(&id, sizeof(id), 1, f);
                    std::fclose(f);
                    if (got == 1) break;
                }
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
                    std::fprintf(stderr, "rank %d: no ncclUniqueId in %s after 120 s (did rank 0 start?)\n", w.rank, idf);
                    std::exit(2);
                }
                std::this_thread::sleep_for(std::chrono::milliseconds(20));
            }
        }
    }
    EQF_NCCL(ncclCommInitRank(&w.comm, w.world, id, w.rank));
    return w;
}

// every rank arrives, then returns: an all-reduce of one int on the world's stream + a wait for it
inline void barrier(World& w, int* dScratch) {
    EQF_NCCL(ncclAllReduce(dScratch, dScratch, 1, ncclInt, ncclSum, w.comm, w.stream));
    EQF_HIP(hipStreamSynchronize(w.stream));
}

// eqf_vio/EQVIO_config_template.yaml:1-29 with the bench overrides of SURVEY.md 8(d) (outlier gate off, fastRiccati = false): the same
// numbers as eqf_vio_amd/synth.py: template_settings_dict(), so that a C++ host and the Python binding can be compared on one stream
inline void template_settings(eqf_settings* s) {
    eqf_settings_default(s);
    s->initialGravityVariance = 1.0;
    s->initialVelocityVariance = 1.0;
    s->initialPointVariance = 5000.0;
    s->biasOmegaProcessVariance = 0.0001;
    s->biasAccelProcessVariance = 0.0001;
    s->gravityProcessVariance = 0.01;
    s->velocityProcessVariance = 0.1;
    s->pointProcessVariance = 0.001;
    s->measurementVariance = 0.003;
    s->velOmegaVariance = 0.0001;
    s->velAccelVariance = 0.0001;
    s->initialBiasOmegaVariance = 1.0;
    s->initialBiasAccelVariance = 1.0;
    s->initialSceneDepth = 1.0;
    s->outlierThreshold = 1e9;
    s->fastRiccati = 0;
    s->useInnovationLift = s->useDiscreteInnovationLift = s->useDiscreteVelocityLift = 1;
    const double x[3] = {-0.0216401454975, -0.064676986768, 0.00981073058949};  // EuRoC cam0, yaml order "xw"
    const double q[4] = {0.7123014606690344, -0.007707179755538301, 0.010499323370588468, 0.7017528002920512};
    for (int i = 0; i < 3; ++i) s->cameraOffset_x[i] = x[i];
    for (int i = 0; i < 4; ++i) s->cameraOffset_q[i] = q[i];
}

inline void finish(World& w) {
    if (w.comm) ncclCommDestroy(w.comm);
    if (w.stream) (void)hipStreamDestroy(w.stream);
}

}  // namespace eqf_rccl
