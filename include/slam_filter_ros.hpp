// slam_filter_ros.hpp — the batched MI355X EKF / UKF behind the reference's OWN abstract class, signature for signature.
//
// Compile this header INSIDE the reference's localization_pkg (it needs what that package already has: ROS, yaml-cpp, Eigen and the
// generated base_pkg messages; none of them exist in the build image of this repository, so this file is checked here only for its
// C-ABI calls, by tests/test_host_driver_cpu.py against stand-in declarations of the few ROS / Eigen / YAML names it touches - see
// INTEGRATION.md §2).  It derives from `::Filter` of
//     ekf_ws/src/localization_pkg/include/localization_pkg/filter.h:54-77
// and overrides exactly its virtuals:
//     readParams(YAML::Node)                              filter.h:59   (EKF::readParams -> readCommonParams, filter.h:105-121)
//     init(float, float, float)                           filter.h:60   (ekf.cpp:29-34, ukf.cpp:29-36)
//     update(Command::ConstPtr, Float32MultiArray::ConstPtr)   filter.h:61   (ekf.cpp:37-179, ukf.cpp:161-372)
//     setupStatePublisher(ros::NodeHandle)                filter.h:65   (ekf.cpp:186-189, ukf.cpp:55-58)
//     publishState()                                      filter.h:66   (ekf.cpp:191-218, ukf.cpp:60-104)
//     Eigen::VectorXd getStateVector()                    filter.h:76   (ekf.cpp:181-184)
// so that localization_node.cpp needs ONE added branch in its factory (localization_node.cpp:33-46) and nothing else: lines 47
// (`filter->readParams(config)`), 127 (`updateNaiveVehPoseEstimate(filter_secondary->getStateVector(), ...)`), 131
// (`filter->update(cmdMsg, lmMeasMsg)`) and 187 (`filter->setupStatePublisher(node)`) compile against it unchanged.
//
// What a batch means behind a one-filter interface: the node feeds ONE command / measurement stream; every instance of the batch
// receives that message (Monte-Carlo replicas of one robot differ by what they are given through the batch entry points below, or
// by nothing at all), instance `published` (default 0) is what publishState() sends and getStateVector() returns.  The per-instance
// entry points (updateBatch, updateSim, errorStats) are additional public members, as in include/slam_filter.hpp.
#pragma once
#ifndef SLAM_AMD_USE_ROS_MSGS
#error "slam_filter_ros.hpp is the adapter for a build inside the reference's localization_pkg: define SLAM_AMD_USE_ROS_MSGS and include localization_pkg/filter.h first (INTEGRATION.md 2); ROS-free hosts use slam_filter.hpp"
#endif

#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "slam_batch.h"

namespace slam_amd {

inline void ros_check(int rc) {
    if (rc != SLAM_OK) throw std::runtime_error(std::string("slam_batch: ") + slam_last_error());   // the reference's error channel (filter.h:5)
}

// slam_config from the YAML::Node the node has already loaded (localization_node.cpp:29-30): the keys readCommonParams reads
// (filter.h:105-121) plus the ones the device-side measurement generator needs; a missing key keeps the committed default.
inline slam_config config_from_yaml(const YAML::Node& config) {
    slam_config c;
    ros_check(slam_config_default(&c));
    auto rd = [](const YAML::Node& n, auto& out) { if (n) out = n.as<std::decay_t<decltype(out)>>(); };
    if (const YAML::Node pn = config["process_noise"]) {
        rd(pn["mean"]["v_d"], c.v_d); rd(pn["mean"]["v_th"], c.v_th); rd(pn["cov"]["V_00"], c.V_00); rd(pn["cov"]["V_11"], c.V_11);
    }
    if (const YAML::Node sn = config["sensing_noise"]) {
        rd(sn["mean"]["w_r"], c.w_r); rd(sn["mean"]["w_b"], c.w_b); rd(sn["cov"]["W_00"], c.W_00); rd(sn["cov"]["W_11"], c.W_11);
    }
    if (const YAML::Node cs = config["constraints"]) {
        if (cs["measurements"]["landmark_id_is_known"]) c.landmark_id_is_known = cs["measurements"]["landmark_id_is_known"].as<bool>() ? 1 : 0;
        rd(cs["measurements"]["min_landmark_separation"], c.min_landmark_separation);
        rd(cs["commands"]["d_max"], c.d_max); rd(cs["commands"]["th_max"], c.th_max);
        rd(cs["vision"]["range_max"], c.range_max); rd(cs["vision"]["fov_min"], c.fov_min); rd(cs["vision"]["fov_max"], c.fov_max);
    }
    if (const YAML::Node ip = config["init_pose"]) { rd(ip["x"], c.init_x); rd(ip["y"], c.init_y); rd(ip["yaw"], c.init_yaw); }
    return c;   // replicate_vw_quirk stays 1: filter.h:116-117 store W_00, W_11 into V and leave W = I2
}

// Common part of the two adapters: handle lifetime, the one-message-for-all update, the batch entry points.
class BatchedFilterRos : public ::Filter {
public:
    BatchedFilterRos(int filter_kind, int batch, int L_max, int device) : kind_(filter_kind), batch_(batch), L_max_(L_max), device_(device) {}
    ~BatchedFilterRos() override { if (h_) slam_destroy(h_); }
    BatchedFilterRos(const BatchedFilterRos&) = delete;
    BatchedFilterRos& operator=(const BatchedFilterRos&) = delete;

    void readParams(YAML::Node config) override {   // filter.h:59
        const slam_config c = config_from_yaml(config);
        if (h_) { slam_destroy(h_); h_ = nullptr; }
        const int kind = (kind_ == SLAM_UKF_SLAM && this->type == FilterChoice::UKF_LOC) ? SLAM_UKF_LOC : kind_;   // localization_node.cpp:40
        ros_check(slam_create(&c, kind, batch_, L_max_, SLAM_F64, device_, &h_));
    }
    void init(float x_0, float y_0, float yaw_0) override {   // filter.h:60
        need();
        ros_check(slam_init(h_, x_0, y_0, yaw_0));
        this->isInit = true;
    }
    // filter.h:61.  lmMeasMsg->data = [id, range, bearing] * k (ekf.cpp:65-76); every instance of the batch gets the message.
    void update(base_pkg::Command::ConstPtr cmdMsg, std_msgs::Float32MultiArray::ConstPtr lmMeasMsg) override {
        need();
        if (kind() == SLAM_UKF_LOC && !map_sent_ && !this->map.empty()) sendTrueMap();   // trueMapCallback filled Filter::map (localization_node.cpp:152-156)
        const int k = (int)(lmMeasMsg->data.size() / 3);
        const int ks = k > 0 ? k : 1;
        meas_.assign((size_t)batch_ * ks * 3, 0.f);
        cnt_.assign((size_t)batch_, k);
        for (int b = 0; b < batch_; ++b)
            for (int i = 0; i < 3 * k; ++i) meas_[(size_t)b * ks * 3 + i] = lmMeasMsg->data[i];
        const float cmd[2] = {(float)cmdMsg->fwd, (float)cmdMsg->ang};
        ros_check(slam_step(h_, cmd, meas_.data(), cnt_.data(), ks));
    }
    Eigen::VectorXd getStateVector() override {   // filter.h:76 (ekf.cpp:181-184); also refreshes lm_IDs, which localization_node.cpp:127 reads next
        need();
        std::vector<double> x((size_t)slam_state_dim_max(h_));
        std::vector<int32_t> ids((size_t)L_max_ + 1);
        int32_t M = 0;
        ros_check(slam_get_state(h_, published, x.data(), nullptr, &M, ids.data(), nullptr));
        this->lm_IDs.assign(ids.begin(), ids.begin() + M);
        const int n = stateDim(M);
        Eigen::VectorXd v(n);
        for (int i = 0; i < n; ++i) v(i) = x[(size_t)i];
        return v;
    }

    // ---- the batch behind the interface ----
    int published = 0;                      // instance publishState() / getStateVector() report
    slam_handle* handle() { need(); return h_; }
    void updateBatch(const float cmd[2], const float* meas, const int32_t* meas_count, int k_stride) { need(); ros_check(slam_step(h_, cmd, meas, meas_count, k_stride)); }
    void setMap(const std::vector<double>& map_xy) { need(); ros_check(slam_set_map(h_, map_xy.data(), (int)(map_xy.size() / 2))); }
    void setSeed(uint64_t seed) { need(); ros_check(slam_set_seed(h_, seed)); }
    void updateSim(const float cmd[2]) { need(); ros_check(slam_step_sim(h_, cmd)); }   // device-side get_cmd (sim_node.py:209-250) + update
    std::vector<double> errorStats() { need(); std::vector<double> e((size_t)batch_); ros_check(slam_error_stats(h_, e.data())); return e; }
    std::vector<int32_t> status() { need(); std::vector<int32_t> f((size_t)batch_); ros_check(slam_status(h_, f.data())); return f; }

protected:
    virtual int stateDim(int M) const = 0;
    int kind() const { return (kind_ == SLAM_UKF_SLAM && this->type == FilterChoice::UKF_LOC) ? SLAM_UKF_LOC : kind_; }
    void need() const { if (!h_) throw std::runtime_error("readParams() has not been called"); }
    void sendTrueMap() {   // Filter::map = [id, x, y] float32 triplets (localization_node.cpp:152-156)
        std::vector<double> xy;
        for (size_t i = 0; i + 2 < this->map.size(); i += 3) { xy.push_back(this->map[i + 1]); xy.push_back(this->map[i + 2]); }
        ros_check(slam_set_map(h_, xy.data(), (int)(xy.size() / 2)));
        map_sent_ = true;
    }
    int kind_, batch_, L_max_, device_;
    slam_handle* h_ = nullptr;
    bool map_sent_ = false;
    std::vector<float> meas_;
    std::vector<int32_t> cnt_;
};

// Drop-in for `std::make_unique<EKF>()` (localization_node.cpp:34).
class BatchedEKFRos : public BatchedFilterRos {
public:
    explicit BatchedEKFRos(int batch = 1, int L_max = 50, int device = 0) : BatchedFilterRos(SLAM_EKF_SLAM, batch, L_max, device) {
        this->type = FilterChoice::EKF_SLAM;   // filter.h:151
    }
    void setupStatePublisher(ros::NodeHandle node) override {   // ekf.cpp:186-189
        this->statePub = node.advertise<base_pkg::EKFState>("/state/ekf", 1);
        need();
        ros_check(slam_track_instance(h_, published));   // publishState() every tick without running the batch's queue
    }
    void publishState() override {   // ekf.cpp:191-218
        need();
        const int nmax = slam_state_dim_max(h_);
        std::vector<double> x((size_t)nmax), P((size_t)nmax * nmax);
        std::vector<int32_t> ids((size_t)L_max_ + 1);
        int32_t M = 0, ts = 0;
        ros_check(slam_get_state(h_, published, x.data(), P.data(), &M, ids.data(), &ts));
        base_pkg::EKFState stateMsg;
        stateMsg.timestep = ts;
        stateMsg.x_v = x[0]; stateMsg.y_v = x[1]; stateMsg.yaw_v = x[2];
        stateMsg.M = M;
        for (int i = 0; i < M; ++i) {   // [id, x, y] triplets (ekf.cpp:203-208)
            stateMsg.landmarks.push_back((float)ids[(size_t)i]);
            stateMsg.landmarks.push_back((float)x[(size_t)(3 + 2 * i)]);
            stateMsg.landmarks.push_back((float)x[(size_t)(4 + 2 * i)]);
        }
        const int n = 3 + 2 * M;
        stateMsg.P.reserve((size_t)n * n);
        for (int i = 0; i < n * n; ++i) stateMsg.P.push_back((float)P[(size_t)i]);   // rows side by side (ekf.cpp:210-216); slam_get_state packs n x n
        this->statePub.publish(stateMsg);
    }
protected:
    int stateDim(int M) const override { return 3 + 2 * M; }
};

// Drop-in for `std::make_unique<UKF>()` (localization_node.cpp:36-40; set `type = FilterChoice::UKF_LOC` BEFORE readParams for ukf_loc,
// as the node does).
class BatchedUKFRos : public BatchedFilterRos {
public:
    explicit BatchedUKFRos(int batch = 1, int L_max = 20, int device = 0) : BatchedFilterRos(SLAM_UKF_SLAM, batch, L_max, device) {
        this->type = FilterChoice::UKF_SLAM;   // filter.h:182
    }
    void setupStatePublisher(ros::NodeHandle node) override {   // ukf.cpp:55-58
        this->statePub = node.advertise<base_pkg::UKFState>("/state/ukf", 1);
    }
    void publishState() override {   // ukf.cpp:60-104
        need();
        const int nmax = slam_state_dim_max(h_);
        std::vector<double> x((size_t)nmax), P((size_t)nmax * nmax);
        std::vector<int32_t> ids((size_t)L_max_ + 1);
        int32_t M = 0, ts = 0;
        ros_check(slam_get_state(h_, published, x.data(), P.data(), &M, ids.data(), &ts));
        const int n = 4 + 2 * M;
        base_pkg::UKFState stateMsg;
        stateMsg.timestep = ts;
        stateMsg.x_v = x[0]; stateMsg.y_v = x[1];
        stateMsg.yaw_v = std::remainder(std::atan2(x[3], x[2]), 2 * 3.14159265358979323846);   // the state holds (cos, sin) of the heading (ukf.cpp:71)
        stateMsg.M = M;
        for (int i = 0; i < M; ++i) {
            stateMsg.landmarks.push_back((float)ids[(size_t)i]);
            stateMsg.landmarks.push_back((float)x[(size_t)(4 + 2 * i)]);
            stateMsg.landmarks.push_back((float)x[(size_t)(5 + 2 * i)]);
        }
        for (int i = 0; i < n * n; ++i) stateMsg.P.push_back((float)P[(size_t)i]);
        std::vector<double> X((size_t)nmax * (2 * nmax + 1));   // sigma points of the last prediction stage, column by column (ukf.cpp:90-101)
        int32_t rows = 0, cols = 0;
        ros_check(slam_get_sigma_points(h_, published, X.data(), &rows, &cols));
        for (size_t i = 0; i < (size_t)rows * cols; ++i) stateMsg.X.push_back((float)X[i]);
        this->statePub.publish(stateMsg);
    }
    Eigen::VectorXd getStateVector() override {   // (x, y, yaw, landmarks): UKF::getStateVector throws for M > 0 (ukf.cpp:48-51); not replicated
        Eigen::VectorXd s = BatchedFilterRos::getStateVector();
        Eigen::VectorXd v(s.size() - 1);
        v(0) = s(0); v(1) = s(1); v(2) = std::atan2(s(3), s(2));
        for (int i = 4; i < s.size(); ++i) v(i - 1) = s(i);
        return v;
    }
protected:
    int stateDim(int M) const override { return 4 + 2 * M; }
};

}  // namespace slam_amd
