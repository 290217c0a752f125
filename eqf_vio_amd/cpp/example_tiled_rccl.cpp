// BASELINE configs[4] from C++: ONE filter whose Sigma is 2-D block-partitioned over the GPUs of a node, one process
// per GPU, the host loop behind the C ABI (eqf_tf_*, csrc/eqf_tiledf.hip) and the ONE thing it needs from its host -- a broadcast of device
// memory along a process row / column -- answered with ncclBroadcast.  Every rank runs the reference's event loop
// (eqf_vio/src/main.cpp:111-170) on the same inputs.
//
//   eqf_example_tiled_rccl --spawn 8 [N=4000] [frames=3] [block=250] [option=value ...]      (grid: 1 -> 1x1, 2 -> 1x2, 4 -> 2x2, 8 -> 2x4)
//   eqf_example_tiled_rccl [N] [frames] [block]                                               (one rank: 1 x 1 grid)
//
// Communicators: the world's, split (ncclCommSplit) into one per process row (color = pr, key = rank: the index inside the group is the
// process COLUMN) and one per process column (color = pc: index = process ROW) -- twice, one set per chain of an update, for a host that
// switches the interleaved chains on (option overlap_chains=1; off by default on a grid, include/eqf_vio_amd.h).  Before the filter starts
// every communicator broadcasts a known pattern from each of its members through the very callback the library will use, and the receivers
// check it: a wrong root convention or a broken split stops here with a message, not as a wrong covariance later.
#include <algorithm>
#include <cmath>

#include "VIOFilterTiled.h"
#include "rccl_host.h"

using namespace eqf_rccl;
using namespace eqf_vio_amd;

namespace {

struct Comms {
    ncclComm_t row[2] = {nullptr, nullptr}, col[2] = {nullptr, nullptr}, all = nullptr;
    long long calls = 0;
    size_t bytes = 0;
};

// eqf_tf_comm::bcast (include/eqf_vio_amd.h): group 0 = my process row (root = process column of the sender), 1 = my process column (root =
// process row), 2 = everybody (root = rank); ordered on `stream`
int bcast(void* ctx, int group, int chain, int root, void* buf, size_t bytes, void* stream) {
    auto* c = static_cast<Comms*>(ctx);
    ncclComm_t comm = group == 0 ? c->row[chain & 1] : (group == 1 ? c->col[chain & 1] : c->all);
    c->calls += 1;
    c->bytes += bytes;
    return ncclBroadcast(buf, buf, bytes, ncclChar, root, comm, static_cast<hipStream_t>(stream)) == ncclSuccess ? 0 : 1;
}

}  // namespace

int main(int argc, char** argv) {
    const int spawned = maybe_spawn(argc, argv);
    if (spawned >= 0) return spawned;
    std::vector<std::string> pos, opts;
    for (int i = 1; i < argc; ++i) (std::string(argv[i]).find('=') != std::string::npos ? opts : pos).push_back(argv[i]);
    const int N = pos.size() > 0 ? std::atoi(pos[0].c_str()) : 4000;
    const int frames = pos.size() > 1 ? std::atoi(pos[1].c_str()) : 3;
    const int bl = pos.size() > 2 ? std::atoi(pos[2].c_str()) : 250;

    World w = init_world();
    int Pr = 1, Pc = w.world;
    if (w.world == 4) Pr = 2, Pc = 2;
    else if (w.world == 8) Pr = 2, Pc = 4;
    else if (w.world != 1 && w.world != 2) {
        std::fprintf(stderr, "rank %d: grids are defined for 1, 2, 4 or 8 ranks (Pr | Pc), not %d\n", w.rank, w.world);
        return 2;
    }
    const int pr = w.rank / Pc, pc = w.rank % Pc;
    Comms comms;
    comms.all = w.comm;
    for (int chain = 0; chain < 2; ++chain) {
        EQF_NCCL(ncclCommSplit(w.comm, pr, w.rank, &comms.row[chain], nullptr));
        EQF_NCCL(ncclCommSplit(w.comm, pc, w.rank, &comms.col[chain], nullptr));
    }

    // ---- the callback, checked before the filter depends on it: every member of every group sends a pattern, the others verify it
    {
        const int cnt = 1024;
        double* d = nullptr;
        EQF_HIP(hipMalloc(&d, cnt * sizeof(double)));
        std::vector<double> h(cnt);
        int bad = 0;
        for (int chain = 0; chain < 2; ++chain)
            for (int group = 0; group < 3; ++group) {
                const int members = group == 0 ? Pc : (group == 1 ? Pr : w.world);
                const int me = group == 0 ? pc : (group == 1 ? pr : w.rank);
                for (int root = 0; root < members; ++root) {
                    // the global rank of `root` inside my group, as the library counts it
                    const int src = group == 0 ? pr * Pc + root : (group == 1 ? root * Pc + pc : root);
                    for (int i = 0; i < cnt; ++i) h[i] = me == root ? 1000.0 * src + 10.0 * group + chain + 1e-3 * i : -1.0;
                    EQF_HIP(hipMemcpyAsync(d, h.data(), cnt * sizeof(double), hipMemcpyHostToDevice, w.stream));
                    if (bcast(&comms, group, chain, root, d, cnt * sizeof(double), w.stream) != 0) bad |= 1;
                    EQF_HIP(hipMemcpyAsync(h.data(), d, cnt * sizeof(double), hipMemcpyDeviceToHost, w.stream));
                    EQF_HIP(hipStreamSynchronize(w.stream));
                    for (int i = 0; i < cnt; ++i)
                        if (h[i] != 1000.0 * src + 10.0 * group + chain + 1e-3 * i) bad |= 2;
                }
            }
        (void)hipFree(d);
        if (bad) {
            std::fprintf(stderr, "rank %d: the broadcast callback failed its check (%d): wrong root convention or communicator split\n", w.rank, bad);
            return 7;
        }
        comms.calls = 0;
        comms.bytes = 0;
    }

    VIOFilter::Settings s;
    template_settings(&s);
    eqf_tf_comm comm{&comms, bcast};
    int exitCode = 0;
    try {
        VIOFilterTiled filter(s, N, bl, Pr, Pc, w.rank, w.device, w.world > 1 ? &comm : nullptr);
        eqf_tf_set_option(filter.handle(), "check_every", 0);
        for (const std::string& kv : opts) {
            const size_t eq = kv.find('=');
            if (eqf_tf_set_option(filter.handle(), kv.substr(0, eq).c_str(), std::atoi(kv.c_str() + eq + 1)) != EQF_OK)
                std::fprintf(stderr, "rank %d: unknown option %s\n", w.rank, kv.c_str());
        }
        std::vector<Vector3d> lm(N);
        for (int i = 0; i < N; ++i) lm[i] = {2 * std::sin(1.3 * i), 2 * std::cos(0.7 * i), 5 + std::sin(0.37 * i)};
        IMUVelocity imu;
        int* dOne = nullptr;
        EQF_HIP(hipMalloc(&dOne, sizeof(int)));
        EQF_HIP(hipMemset(dOne, 0, sizeof(int)));
        int k = 0;
        long steps = 0;
        std::chrono::steady_clock::time_point t0;
        for (int f = 0; f < frames; ++f) {
            if (f == 1) {  // (the first frame allocates and initialises: not timed)
                filter.synchronize();
                barrier(w, dOne);
                t0 = std::chrono::steady_clock::now();
                steps = 0;
            }
            VisionMeasurement meas;
            meas.stamp = 0.05 * f + 0.0025;
            for (; 0.005 * k < meas.stamp; ++k, ++steps) {  // main.cpp:113
                imu.stamp = 0.005 * k;
                imu.omega = {0.02 * std::sin(0.015 * k), 0.015 * std::cos(0.01 * k), 0.01 * std::sin(0.0075 * k)};
                imu.accel = {GRAVITY_CONSTANT + 0.05 * std::sin(0.02 * k), 0.04 * std::cos(0.015 * k), 0.03 * std::sin(0.0125 * k)};  // body x up
                filter.processIMUData(imu);
            }
            for (int i = 0; i < N; ++i) {
                if ((f + i) % 97 == 0) continue;  // out of view on this frame: landmarks leave and come back (VIOFilter.cpp:345-443 on slots)
                const double n = std::sqrt(lm[i][0] * lm[i][0] + lm[i][1] * lm[i][1] + lm[i][2] * lm[i][2]);
                Point3d b;
                b.p = {lm[i][0] / n, lm[i][1] / n, lm[i][2] / n};
                b.id = i;
                meas.bearings.push_back(b);
            }
            meas.numberOfBearings = int(meas.bearings.size());
            filter.processVisionData(meas);
            ++steps;
        }
        filter.synchronize();
        barrier(w, dOne);
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (eqf_tf_check(filter.handle()) != EQF_OK || eqf_tf_device_error(filter.handle()) != 0) {
            std::fprintf(stderr, "rank %d: %s (device flag %d)\n", w.rank, eqf_tf_last_error(filter.handle()), eqf_tf_device_error(filter.handle()));
            exitCode = 6;
        }
        const VIOState est = filter.stateEstimate();  // replicated: the same on every rank
        // |Sigma|_F from the partitioned matrix: the replicated base rows on rank 0 + every rank's local blocks, one all-reduce -- through
        // stateCovariance() when the matrix is small enough to gather (collective; every rank calls it)
        double fro = -1.0;
        if (N <= 1000) {
            const MatrixXd S = filter.stateCovariance();
            fro = 0;
            for (double v : S.data) fro += v * v;
            fro = std::sqrt(fro);
        }
        // every rank must hold the same replicated state: compare the pose with rank 0's
        double pose[8] = {est.pose.R.w, est.pose.R.x, est.pose.R.y, est.pose.R.z, est.pose.x[0], est.pose.x[1], est.pose.x[2], double(est.bodyLandmarks.size())};
        double* dPose = nullptr;
        EQF_HIP(hipMalloc(&dPose, sizeof(pose)));
        EQF_HIP(hipMemcpy(dPose, pose, sizeof(pose), hipMemcpyHostToDevice));
        EQF_NCCL(ncclBroadcast(dPose, dPose, 8, ncclDouble, 0, w.comm, w.stream));
        EQF_HIP(hipStreamSynchronize(w.stream));
        double pose0[8];
        EQF_HIP(hipMemcpy(pose0, dPose, sizeof(pose0), hipMemcpyDeviceToHost));
        for (int i = 0; i < 8; ++i)
            if (pose0[i] != pose[i]) {
                std::fprintf(stderr, "rank %d: replicated state differs from rank 0's (component %d: %.17g vs %.17g)\n", w.rank, i, pose[i], pose0[i]);
                exitCode = 8;
            }
        if (w.rank == 0) {
            std::printf("t=%.4f N=%zu pos=(%.9f %.9f %.9f) q=(%.9f %.9f %.9f %.9f) |Sigma|_F=%.9e\n", filter.getTime(), est.bodyLandmarks.size(), est.pose.x[0],
                est.pose.x[1], est.pose.x[2], est.pose.R.w, est.pose.R.x, est.pose.R.y, est.pose.R.z, fro);
            const int timedFrames = std::max(frames - 1, 1);
            std::printf("{\"host\": \"C++ + RCCL\", \"n_gpus\": %d, \"grid\": \"%d x %d\", \"landmarks\": %d, \"block_landmarks\": %d, \"steps\": %ld, \"value\": %.2f, "
                        "\"unit\": \"steps/s\", \"ms_per_frame\": %.3f, \"broadcasts_per_frame\": %.1f, \"broadcast_MB_per_frame_this_rank\": %.2f}\n",
                w.world, Pr, Pc, N, bl, steps, frames > 1 ? double(steps) / dt : 0.0, frames > 1 ? dt * 1e3 / timedFrames : 0.0,
                double(comms.calls) / frames, double(comms.bytes) / frames / 1e6);
        }
        (void)hipFree(dOne);
        (void)hipFree(dPose);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "rank %d: exception: %s\n", w.rank, e.what());
        exitCode = 3;
    }
    for (int chain = 0; chain < 2; ++chain) {
        if (comms.row[chain]) ncclCommDestroy(comms.row[chain]);
        if (comms.col[chain]) ncclCommDestroy(comms.col[chain]);
    }
    finish(w);
    return exitCode;
}
