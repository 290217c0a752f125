// Minimal C++ caller of the facade, shaped like the reference's offline runner (eqf_vio/src/main.cpp:111-170):
// events are interleaved by "imu.stamp < meas.stamp", the state is read after every vision call.
// Usage: eqf_example <N landmarks> <frames> [aux | level | init]  -- runs a small synthetic sequence and prints the final
// pose and |Sigma|_F.  With "aux" the filter starts from AuxiliaryFilterData + setInertialPoints (VIOFilter.cpp:51-58,
// 74-118) instead of the gravity alignment at the first IMU sample; with "init" from an explicit initialiseFromIMUData
// call (VIOFilter.cpp:133-144; same result as the lazy one).  With "level" the vehicle rests level: the reference's gravity
// chart is then singular and its first Riccati step throws std::domain_error (SO3.cpp:160-161) -- here the device raises
// its sticky flag and the facade throws the same exception; the example reports it and exits with status 3.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "VIOFilter.h"

using namespace eqf_vio_amd;

int main(int argc, char** argv) {
    const int N = argc > 1 ? std::atoi(argv[1]) : 20;
    const int frames = argc > 2 ? std::atoi(argv[2]) : 10;
    VIOFilter::Settings s;
    s.initialPointVariance = 5000.0;  // eqf_vio/EQVIO_config_template.yaml values
    s.measurementVariance = 0.003;
    s.velOmegaVariance = s.velAccelVariance = 1e-4;
    s.outlierThreshold = 1e9;
    const bool aux = argc > 3 && std::string(argv[3]) == "aux";
    const bool level = argc > 3 && std::string(argv[3]) == "level";
    const bool init = argc > 3 && std::string(argv[3]) == "init";
    std::vector<Vector3d> lm(N);
    for (int i = 0; i < N; ++i) lm[i] = {2 * std::sin(1.3 * i), 2 * std::cos(0.7 * i), 5 + std::sin(0.37 * i)};
    AuxiliaryFilterData ad;
    ad.initialAttitude = {std::sqrt(0.5), 0, -std::sqrt(0.5), 0};  // body x = inertial up
    ad.initialPosition = {0.3, -0.2, 1.0};
    ad.cameraOffset.R = {0.98, 0.1, -0.1, std::sqrt(1 - 0.98 * 0.98 - 0.02)};
    ad.cameraOffset.x = {0.1, -0.05, 0.02};
    VIOFilter filter = aux ? VIOFilter(ad, s, N) : VIOFilter(s, N);
    if (aux) {
        // inertial position of landmark i = pose * cameraOffset * (camera-frame point)
        auto rot = [](const Quaterniond& q, const Vector3d& v) {
            const Vector3d u = {q.x, q.y, q.z};
            const Vector3d t = {2 * (u[1] * v[2] - u[2] * v[1]), 2 * (u[2] * v[0] - u[0] * v[2]), 2 * (u[0] * v[1] - u[1] * v[0])};
            return Vector3d{v[0] + q.w * t[0] + u[1] * t[2] - u[2] * t[1], v[1] + q.w * t[1] + u[2] * t[0] - u[0] * t[2],
                v[2] + q.w * t[2] + u[0] * t[1] - u[1] * t[0]};
        };
        std::vector<Point3d> pts(N);
        for (int i = 0; i < N; ++i) {
            const Vector3d b = rot(ad.cameraOffset.R, lm[i]);
            const Vector3d w = rot(ad.initialAttitude, {b[0] + ad.cameraOffset.x[0], b[1] + ad.cameraOffset.x[1], b[2] + ad.cameraOffset.x[2]});
            pts[i].p = {w[0] + ad.initialPosition[0], w[1] + ad.initialPosition[1], w[2] + ad.initialPosition[2]};
            pts[i].id = 100 + 2 * i;
        }
        filter.setInertialPoints(pts);
    }
    // vehicle at rest, tilted so that body x is "up" (a level start makes the reference's gravity chart singular)
    IMUVelocity imu;
    imu.accel = {GRAVITY_CONSTANT, 0, 0};
    if (level) imu.accel = {0, 0, GRAVITY_CONSTANT};
    if (init) filter.initialiseFromIMUData(imu);
    int k = 0;
    try {
    for (int f = 0; f < frames; ++f) {
        VisionMeasurement meas;
        meas.stamp = 0.05 * f + 0.0025;
        for (; 0.005 * k < meas.stamp; ++k) {  // main.cpp:113
            imu.stamp = 0.005 * k;
            filter.processIMUData(imu);
        }
        meas.numberOfBearings = N;
        meas.bearings.resize(N);
        for (int i = 0; i < N; ++i) {
            const double n = std::sqrt(lm[i][0] * lm[i][0] + lm[i][1] * lm[i][1] + lm[i][2] * lm[i][2]);
            meas.bearings[i].p = {lm[i][0] / n, lm[i][1] / n, lm[i][2] / n};
            meas.bearings[i].id = aux ? 100 + 2 * i : i;
        }
        filter.processVisionData(meas);
        const VIOState est = filter.stateEstimate();
        if (f == frames - 1) {
            const MatrixXd S = filter.stateCovariance();
            double fro = 0;
            for (double v : S.data) fro += v * v;
            std::printf("t=%.4f N=%zu pos=(%.6f %.6f %.6f) q=(%.6f %.6f %.6f %.6f) |Sigma|_F=%.6e\n", filter.getTime(),
                est.bodyLandmarks.size(), est.pose.x[0], est.pose.x[1], est.pose.x[2], est.pose.R.w, est.pose.R.x, est.pose.R.y,
                est.pose.R.z, std::sqrt(fro));
        }
    }
    } catch (const std::domain_error& e) {
        std::printf("std::domain_error: %s\n", e.what());
        return 3;
    }
    return 0;
}
