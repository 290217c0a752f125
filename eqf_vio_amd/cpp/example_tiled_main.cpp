// The partitioned filter from plain C++ host code (g++, no HIP headers): VIOFilterTiled on a 1 x 1 grid -- BASELINE configs[4]'s path with
// one rank, no exchange callback needed -- driven like the reference's offline runner (eqf_vio/src/main.cpp:111-170): events interleaved
// by "imu.stamp < meas.stamp", the state read after every vision call.  Landmark i is in view on frames with (f + i) % 7 != 0, so
// landmarks leave and come back (VIOFilter.cpp:345-443 on slots).
// Usage: eqf_example_tiled <N landmarks> <frames> <block landmarks> [timing | host | -] [option=value ...]
//   prints the final pose, |Sigma|_F and the landmark count; with "timing" also the wall time per frame and the time the host spends in a
//   processVisionData call (the call returns when everything is ENQUEUED; the pivot check is switched off, so nothing synchronises -- but a
//   host that runs a whole frame ahead of the device waits for room in the device's queues); with "host" the device is drained before every
//   vision call, so that the call's time is the host's own cost of an update.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/eqf_vio_amd_debug.h"  // eqf_tf_graph_launches (a measurement hook: this example prints it)
#include "VIOFilterTiled.h"

using namespace eqf_vio_amd;

int main(int argc, char** argv) {
    const int N = argc > 1 ? std::atoi(argv[1]) : 20;
    const int frames = argc > 2 ? std::atoi(argv[2]) : 10;
    const int bl = argc > 3 ? std::atoi(argv[3]) : 8;
    const bool idle = argc > 4 && std::string(argv[4]) == "host";  // "host": the device is idle at every vision call -- the call's time is the host's own
    const bool timing = idle || (argc > 4 && std::string(argv[4]) == "timing");
    VIOFilter::Settings s;
    s.initialPointVariance = 5000.0;  // eqf_vio/EQVIO_config_template.yaml values
    s.measurementVariance = 0.003;
    s.velOmegaVariance = s.velAccelVariance = 1e-4;
    s.outlierThreshold = 1e9;
    std::vector<Vector3d> lm(N);
    for (int i = 0; i < N; ++i) lm[i] = {2 * std::sin(1.3 * i), 2 * std::cos(0.7 * i), 5 + std::sin(0.37 * i)};
    VIOFilterTiled filter(s, N, bl);
    if (timing) eqf_tf_set_option(filter.handle(), "check_every", 0);
    for (int i = 5; i < argc; ++i) {  // options of the host loop (include/eqf_vio_amd.h: eqf_tf_set_option), e.g. graphs=1 panel_ahead=1
        const std::string kv(argv[i]);
        const size_t eq = kv.find('=');
        if (eq != std::string::npos) eqf_tf_set_option(filter.handle(), kv.substr(0, eq).c_str(), std::atoi(kv.c_str() + eq + 1));
    }
    IMUVelocity imu;
    imu.accel = {GRAVITY_CONSTANT, 0, 0};  // at rest, body x up
    int k = 0;
    double hostMs = 0.0;
    std::chrono::steady_clock::time_point t0;
    try {
        for (int f = 0; f < frames; ++f) {
            if (f == 1) {  // (the first frame allocates and initialises: not timed)
                filter.synchronize();
                t0 = std::chrono::steady_clock::now();
            }
            VisionMeasurement meas;
            meas.stamp = 0.05 * f + 0.0025;
            for (; 0.005 * k < meas.stamp; ++k) {  // main.cpp:113
                imu.stamp = 0.005 * k;
                filter.processIMUData(imu);
            }
            for (int i = 0; i < N; ++i) {
                if (!timing && (f + i) % 7 == 0) continue;  // out of view on this frame
                const double n = std::sqrt(lm[i][0] * lm[i][0] + lm[i][1] * lm[i][1] + lm[i][2] * lm[i][2]);
                Point3d b;
                b.p = {lm[i][0] / n, lm[i][1] / n, lm[i][2] / n};
                b.id = i;
                meas.bearings.push_back(b);
            }
            meas.numberOfBearings = int(meas.bearings.size());
            if (idle) filter.synchronize();
            const auto h0 = std::chrono::steady_clock::now();
            filter.processVisionData(meas);
            if (f >= 1) hostMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
        }
        filter.synchronize();
        const double wallMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        const VIOState est = filter.stateEstimate();
        const MatrixXd S = filter.stateCovariance();
        double fro = 0;
        for (double v : S.data) fro += v * v;
        std::printf("t=%.4f N=%zu pos=(%.6f %.6f %.6f) q=(%.6f %.6f %.6f %.6f) |Sigma|_F=%.6e\n", filter.getTime(), est.bodyLandmarks.size(),
            est.pose.x[0], est.pose.x[1], est.pose.x[2], est.pose.R.w, est.pose.R.x, est.pose.R.y, est.pose.R.z, std::sqrt(fro));
        if (timing && frames > 1)
            std::printf("host time per processVisionData call %.3f ms (%s), wall time per frame %.3f ms, %d frames, %lld updates replayed from a hipGraph\n",
                hostMs / (frames - 1), idle ? "enqueue only: the device was idle at every call" : "the host runs ahead of the device until its queues are full",
                wallMs / (frames - 1), frames - 1, eqf_tf_graph_launches(filter.handle()));
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return 3;
    }
    return 0;
}
