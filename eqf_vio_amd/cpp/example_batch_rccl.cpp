// BASELINE configs[3] from C++: a batch of independent filters sharded over the GPUs of one node, one process per GPU, RCCL over xGMI for
// the input scatter and the result gather and for nothing else (north_star: "host code stays C++").  The event loop
// each rank replays is the reference's offline runner (eqf_vio/src/main.cpp:111-170): IMU records while imu.stamp < meas.stamp, then the
// vision frame.  The C++ twin of bench.py's job (eqf_vio_amd/shard.py does the same two exchanges through torch.distributed).
//
//   eqf_example_batch_rccl --spawn 8 [filters_total=64] [N=200] [frames=20] [--streams FILE] [--out FILE]
//   eqf_example_batch_rccl [filters_total] ...            (one rank; or under a launcher that sets EQF_RANK / EQF_WORLD / EQF_ID_FILE)
//
//   rank 0   builds (or reads: --streams) every filter's input stream, packs one slice per rank in the layout of eqf_stream_upload and
//            puts the slices in its HBM
//   scatter  ONE ncclGroupStart/End: rank 0 ncclSend's slice r to rank r, rank r ncclRecv's it (GPU to GPU)
//   replay   eqf_stream_upload, then events by index -- no host -> device traffic in the timed loop; barrier on both sides, max over ranks
//   gather   pose (q, x) and |Sigma|_F per filter: ONE group of ncclSend (every rank) / ncclRecv (rank 0)
//
// --streams FILE: int32 K, F, N, B_total; double imu[K][B_total][7], vstamps[F][B_total]; int32 ids[N]; double bearings[F][B_total][N][3]
// (what tests/test_gpu_rccl_hosts.py writes from the synthetic streams of SURVEY.md 8d, so that the results can be checked against the
// Python binding on the same inputs).  --out FILE: double res[B_total][8] = q (4), x (3), |Sigma|_F.
#include <algorithm>
#include <cmath>
#include <cstdint>

#include "../../include/eqf_vio_amd.h"
#include "rccl_host.h"

using namespace eqf_rccl;

namespace {

struct Streams {  // every filter of the job, filter-major slices are cut out of it per rank
    int K = 0, F = 0, N = 0, B = 0;
    std::vector<double> imu, vst, bear;
    std::vector<int> ids;
};

// A stand-in generator (no file given): the vehicle hovers, tilted so that body x is up, with a slow per-filter oscillation; N fixed
// landmarks in front of the camera.  Deterministic in (filter index, k).  Not the bench's stream (that one is synth.py's) -- enough to
// keep every filter busy with well-posed numbers.
Streams synth(int Btot, int N, int frames) {
    Streams s;
    s.B = Btot;
    s.N = N;
    s.F = frames;
    s.K = 10 * frames + 1;
    s.imu.resize(size_t(s.K) * Btot * 7);
    s.vst.resize(size_t(s.F) * Btot);
    s.bear.resize(size_t(s.F) * Btot * N * 3);
    s.ids.resize(N);
    for (int i = 0; i < N; ++i) s.ids[i] = i;
    for (int k = 0; k < s.K; ++k)
        for (int b = 0; b < Btot; ++b) {
            double* r = &s.imu[(size_t(k) * Btot + b) * 7];
            const double t = 0.005 * k, ph = 0.37 * b;
            r[0] = t;
            r[1] = 0.02 * std::sin(3.0 * t + ph);
            r[2] = 0.015 * std::cos(2.0 * t + ph);
            r[3] = 0.01 * std::sin(1.5 * t + 2 * ph);
            r[4] = 9.81 + 0.05 * std::sin(4.0 * t + ph);
            r[5] = 0.04 * std::cos(3.0 * t + ph);
            r[6] = 0.03 * std::sin(2.5 * t + ph);
        }
    for (int f = 0; f < s.F; ++f)
        for (int b = 0; b < Btot; ++b) {
            s.vst[size_t(f) * Btot + b] = 0.05 * f + 0.0025;
            for (int i = 0; i < N; ++i) {
                double p[3] = {2 * std::sin(1.3 * i + 0.1 * b), 2 * std::cos(0.7 * i - 0.05 * b), 5 + std::sin(0.37 * i)};
                p[0] += 0.002 * f * std::sin(0.2 * b + i);
                p[1] += 0.002 * f * std::cos(0.3 * b + i);
                const double n = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
                double* y = &s.bear[((size_t(f) * Btot + b) * N + i) * 3];
                for (int c = 0; c < 3; ++c) y[c] = p[c] / n;
            }
        }
    return s;
}

bool read_streams(const char* path, Streams& s) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    int32_t h[4];
    bool ok = std::fread(h, sizeof(int32_t), 4, f) == 4;
    if (ok) {
        s.K = h[0], s.F = h[1], s.N = h[2], s.B = h[3];
        s.imu.resize(size_t(s.K) * s.B * 7);
        s.vst.resize(size_t(s.F) * s.B);
        s.ids.resize(s.N);
        s.bear.resize(size_t(s.F) * s.B * s.N * 3);
        std::vector<int32_t> ids(s.N);
        ok = std::fread(s.imu.data(), 8, s.imu.size(), f) == s.imu.size() && std::fread(s.vst.data(), 8, s.vst.size(), f) == s.vst.size() &&
             std::fread(ids.data(), 4, ids.size(), f) == ids.size() && std::fread(s.bear.data(), 8, s.bear.size(), f) == s.bear.size();
        for (int i = 0; i < s.N; ++i) s.ids[i] = ids[i];
    }
    std::fclose(f);
    return ok;
}

// slice of filters [b0, b0 + B) in the layouts of eqf_stream_upload, one contiguous buffer: imu [K][B][7] | vst [F][B] | bear [F][B][N][3]
size_t slice_doubles(int K, int F, int N, int B) { return size_t(K) * B * 7 + size_t(F) * B + size_t(F) * B * N * 3; }
void pack_slice(const Streams& s, int b0, int B, double* dst) {
    double* p = dst;
    for (int k = 0; k < s.K; ++k, p += size_t(B) * 7) std::memcpy(p, &s.imu[(size_t(k) * s.B + b0) * 7], sizeof(double) * B * 7);
    for (int f = 0; f < s.F; ++f, p += B) std::memcpy(p, &s.vst[size_t(f) * s.B + b0], sizeof(double) * B);
    for (int f = 0; f < s.F; ++f, p += size_t(B) * s.N * 3)
        std::memcpy(p, &s.bear[(size_t(f) * s.B + b0) * s.N * 3], sizeof(double) * B * s.N * 3);
}

}  // namespace

int main(int argc, char** argv) {
    const int spawned = maybe_spawn(argc, argv);
    if (spawned >= 0) return spawned;
    std::vector<std::string> pos;
    std::string streamsPath, outPath;
    for (int i = 1; i < argc; ++i) {
        const std::string a(argv[i]);
        if (a == "--streams" && i + 1 < argc) streamsPath = argv[++i];
        else if (a == "--out" && i + 1 < argc) outPath = argv[++i];
        else pos.push_back(a);
    }
    int Btot = pos.size() > 0 ? std::atoi(pos[0].c_str()) : 64;
    int N = pos.size() > 1 ? std::atoi(pos[1].c_str()) : 200;
    int frames = pos.size() > 2 ? std::atoi(pos[2].c_str()) : 20;

    World w = init_world();
    int* dOne = nullptr;
    EQF_HIP(hipMalloc(&dOne, sizeof(int)));
    EQF_HIP(hipMemset(dOne, 0, sizeof(int)));

    // ---- rank 0: the streams of the whole job; the shape goes to everybody first (ncclBroadcast of four ints)
    Streams all;
    int shape[4] = {0, 0, 0, 0};
    if (w.rank == 0) {
        if (!streamsPath.empty()) {
            if (!read_streams(streamsPath.c_str(), all)) {
                std::fprintf(stderr, "rank 0: cannot read %s\n", streamsPath.c_str());
                return 2;
            }
        } else {
            all = synth(Btot, N, frames);
        }
        shape[0] = all.K, shape[1] = all.F, shape[2] = all.N, shape[3] = all.B;
    }
    int* dShape = nullptr;
    EQF_HIP(hipMalloc(&dShape, sizeof(shape)));
    EQF_HIP(hipMemcpy(dShape, shape, sizeof(shape), hipMemcpyHostToDevice));
    EQF_NCCL(ncclBroadcast(dShape, dShape, 4, ncclInt, 0, w.comm, w.stream));
    EQF_HIP(hipStreamSynchronize(w.stream));
    EQF_HIP(hipMemcpy(shape, dShape, sizeof(shape), hipMemcpyDeviceToHost));
    const int K = shape[0], F = shape[1];
    N = shape[2];
    Btot = shape[3];
    if (Btot % w.world != 0) {
        std::fprintf(stderr, "rank %d: %d filters do not divide over %d ranks\n", w.rank, Btot, w.world);
        return 2;
    }
    const int B = Btot / w.world;  // filter b of the job runs on rank b / B (shard.py's rule)
    const size_t nd = slice_doubles(K, F, N, B);

    // ---- scatter: rank 0 holds every slice in HBM; ONE group of point-to-point transfers
    double* dMine = nullptr;
    double* dAll = nullptr;
    int* dIds = nullptr;
    EQF_HIP(hipMalloc(&dMine, nd * sizeof(double)));
    EQF_HIP(hipMalloc(&dIds, sizeof(int) * std::max(N, 1)));
    if (w.rank == 0) {
        std::vector<double> packed(nd * w.world);
        for (int r = 0; r < w.world; ++r) pack_slice(all, r * B, B, &packed[nd * r]);
        EQF_HIP(hipMalloc(&dAll, packed.size() * sizeof(double)));
        EQF_HIP(hipMemcpy(dAll, packed.data(), packed.size() * sizeof(double), hipMemcpyHostToDevice));
        EQF_HIP(hipMemcpy(dIds, all.ids.data(), sizeof(int) * N, hipMemcpyHostToDevice));
    }
    const auto tS0 = std::chrono::steady_clock::now();
    EQF_NCCL(ncclGroupStart());
    if (w.rank == 0)
        for (int r = 1; r < w.world; ++r) EQF_NCCL(ncclSend(dAll + nd * r, nd, ncclDouble, r, w.comm, w.stream));
    else
        EQF_NCCL(ncclRecv(dMine, nd, ncclDouble, 0, w.comm, w.stream));
    EQF_NCCL(ncclGroupEnd());
    if (w.rank == 0) EQF_HIP(hipMemcpyAsync(dMine, dAll, nd * sizeof(double), hipMemcpyDeviceToDevice, w.stream));
    EQF_NCCL(ncclBroadcast(dIds, dIds, N, ncclInt, 0, w.comm, w.stream));
    EQF_HIP(hipStreamSynchronize(w.stream));
    const double scatterMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tS0).count();

    std::vector<double> mine(nd);
    std::vector<int> ids(std::max(N, 1));
    EQF_HIP(hipMemcpy(mine.data(), dMine, nd * sizeof(double), hipMemcpyDeviceToHost));
    EQF_HIP(hipMemcpy(ids.data(), dIds, sizeof(int) * N, hipMemcpyDeviceToHost));
    const double* imu = mine.data();
    const double* vst = imu + size_t(K) * B * 7;
    const double* bear = vst + size_t(F) * B;

    // ---- the rank's filters: one handle, batch = B (one batched launch per step), streams resident in HBM
    eqf_settings st;
    template_settings(&st);
    eqf_filter* fh = nullptr;
    int rc = eqf_create(&st, N, B, w.device, EQF_PRECISION_F64, &fh);
    if (rc != EQF_OK) {
        std::fprintf(stderr, "rank %d: eqf_create failed with status %d (no CPU fallback)\n", w.rank, rc);
        return 4;
    }
    rc = eqf_stream_upload(fh, K, imu, F, vst, N, ids.data(), bear);
    if (rc != EQF_OK) {
        std::fprintf(stderr, "rank %d: eqf_stream_upload failed with status %d\n", w.rank, rc);
        return 4;
    }
    // event schedule of main.cpp:111-170 from filter 0's stamps of this rank (all filters share the schedule)
    auto replay = [&](int f0, int f1, int& k) {
        long steps = 0;
        for (int f = f0; f < f1; ++f) {
            for (; k < K && imu[size_t(k) * B * 7] < vst[size_t(f) * B]; ++k, ++steps) eqf_stream_imu(fh, k);
            eqf_stream_vision(fh, f);
            ++steps;
        }
        return steps;
    };
    int k = 0;
    const int warm = std::min(2, F);  // the first frames add the landmarks: not timed
    replay(0, warm, k);
    eqf_synchronize(fh);
    barrier(w, dOne);
    const auto t0 = std::chrono::steady_clock::now();
    const long steps = replay(warm, F, k);
    eqf_synchronize(fh);
    barrier(w, dOne);
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    double* dT = nullptr;
    EQF_HIP(hipMalloc(&dT, sizeof(double)));
    EQF_HIP(hipMemcpy(dT, &dt, sizeof(double), hipMemcpyHostToDevice));
    EQF_NCCL(ncclAllReduce(dT, dT, 1, ncclDouble, ncclMax, w.comm, w.stream));
    EQF_HIP(hipStreamSynchronize(w.stream));
    EQF_HIP(hipMemcpy(&dt, dT, sizeof(double), hipMemcpyDeviceToHost));

    // ---- results: pose + |Sigma|_F per filter, gathered to rank 0 as ONE group
    const int n = 11 + 3 * N;
    std::vector<double> res(size_t(B) * 8), S(size_t(n) * n), v(3), p(size_t(3) * std::max(N, 1));
    for (int b = 0; b < B; ++b) {
        double* r = &res[size_t(b) * 8];
        eqf_get_state_estimate(fh, b, r, r + 4, v.data(), p.data());
        eqf_get_sigma(fh, b, S.data(), n);
        double fro = 0;
        for (double x : S) fro += x * x;
        r[7] = std::sqrt(fro);
    }
    const int err = eqf_device_error(fh);
    double* dRes = nullptr;
    double* dGather = nullptr;
    EQF_HIP(hipMalloc(&dRes, res.size() * sizeof(double)));
    EQF_HIP(hipMemcpy(dRes, res.data(), res.size() * sizeof(double), hipMemcpyHostToDevice));
    if (w.rank == 0) EQF_HIP(hipMalloc(&dGather, res.size() * sizeof(double) * w.world));
    EQF_NCCL(ncclGroupStart());
    if (w.rank == 0)
        for (int r = 1; r < w.world; ++r) EQF_NCCL(ncclRecv(dGather + res.size() * r, res.size(), ncclDouble, r, w.comm, w.stream));
    else
        EQF_NCCL(ncclSend(dRes, res.size(), ncclDouble, 0, w.comm, w.stream));
    EQF_NCCL(ncclGroupEnd());
    if (w.rank == 0) EQF_HIP(hipMemcpyAsync(dGather, dRes, res.size() * sizeof(double), hipMemcpyDeviceToDevice, w.stream));
    int* dErr = nullptr;
    EQF_HIP(hipMalloc(&dErr, sizeof(int)));
    EQF_HIP(hipMemcpy(dErr, &err, sizeof(int), hipMemcpyHostToDevice));
    EQF_NCCL(ncclAllReduce(dErr, dErr, 1, ncclInt, ncclMax, w.comm, w.stream));
    EQF_HIP(hipStreamSynchronize(w.stream));
    int errAll = 0;
    EQF_HIP(hipMemcpy(&errAll, dErr, sizeof(int), hipMemcpyDeviceToHost));
    int exitCode = errAll ? 6 : 0;
    if (w.rank == 0) {
        std::vector<double> g(res.size() * w.world);
        EQF_HIP(hipMemcpy(g.data(), dGather, g.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int b = 0; b < Btot; ++b) {
            const double* r = &g[size_t(b) * 8];
            if (b < 4 || b == Btot - 1)
                std::printf("filter %d (rank %d): q=(%.9f %.9f %.9f %.9f) x=(%.9f %.9f %.9f) |Sigma|_F=%.9e\n", b, b / B, r[0], r[1], r[2], r[3], r[4],
                    r[5], r[6], r[7]);
            if (!std::isfinite(r[7])) exitCode = 6;
        }
        std::printf("{\"host\": \"C++ + RCCL\", \"n_gpus\": %d, \"filters_total\": %d, \"filters_per_gpu\": %d, \"landmarks\": %d, \"steps\": %ld, "
                    "\"value\": %.1f, \"unit\": \"steps/s\", \"ms_per_step\": %.5f, \"scatter_ms\": %.3f, \"scatter_bytes\": %zu, \"device_error_flag\": %d}\n",
            w.world, Btot, B, N, steps, double(steps) * Btot / dt, dt * 1e3 / double(steps), scatterMs, nd * sizeof(double) * size_t(w.world - 1), errAll);
        if (!outPath.empty()) {
            FILE* f = std::fopen(outPath.c_str(), "wb");
            if (!f || std::fwrite(g.data(), 8, g.size(), f) != g.size()) exitCode = 2;
            if (f) std::fclose(f);
        }
    }
    eqf_destroy(fh);
    for (void* q : {(void*)dOne, (void*)dShape, (void*)dMine, (void*)dAll, (void*)dIds, (void*)dT, (void*)dRes, (void*)dGather, (void*)dErr})
        if (q) (void)hipFree(q);
    finish(w);
    return exitCode;
}
