// slam_filter.hpp — ROS-free C++ mirror of the reference's `Filter` interface over the C ABI (slam_batch.h).
//
// Reference: ekf_ws/src/localization_pkg/include/localization_pkg/filter.h:54-77 (abstract class Filter),
// :148-174 (class EKF); caller ekf_ws/src/localization_pkg/src/localization_node.cpp:33-47 (factory),
// :90-106 (initCallback -> init), :108-140 (iterate -> update, publishState).
// Same method names, argument meaning and error convention (std::runtime_error) as the reference, so the node
// harness can hold a `std::unique_ptr<Filter>` to a BatchedEKF exactly as it holds an EKF today.  The message
// types are shims with the fields the hot path reads (ROS is not available in the build image); on a ROS box
// they are replaced by base_pkg::Command / std_msgs::Float32MultiArray via the two typedefs below
// (INTEGRATION.md §2).
#pragma once
#include <cmath>
#include <cstdint>
#include <memory>
#include <queue>
#include <stdexcept>
#include <string>
#include <vector>

#include "slam_batch.h"
#include "slam_pgs.h"

namespace slam_amd {

#ifndef SLAM_AMD_USE_ROS_MSGS
struct Command {            // base_pkg/Command.msg:3-5
    float fwd = 0.f, ang = 0.f;
    using ConstPtr = std::shared_ptr<const Command>;
};
struct Float32MultiArray {  // std_msgs/Float32MultiArray: data = [id, range, bearing] * k (sim_node.py:245-249)
    std::vector<float> data;
    using ConstPtr = std::shared_ptr<const Float32MultiArray>;
};
#endif

enum class FilterChoice { NOT_SET = 0, EKF_SLAM, UKF_LOC, UKF_SLAM, POSE_GRAPH_SLAM, NAIVE_COMMAND_PROPAGATION };  // filter.h:44-51

// EKFState.msg payload as EKF::publishState fills it (ekf.cpp:192-220)
struct EKFState {
    int32_t timestep = 0;
    float x_v = 0, y_v = 0, yaw_v = 0;
    int32_t M = 0;
    std::vector<float> landmarks;  // [id, x, y] * M   (ekf.cpp:203-208)
    std::vector<float> P;          // row-major (3+2M)^2 (ekf.cpp:211-217)
};

inline void check(int rc) {
    if (rc != SLAM_OK) throw std::runtime_error(std::string("slam_batch: ") + slam_last_error());
}

class Filter {  // filter.h:54-77
public:
    FilterChoice type = FilterChoice::NOT_SET;
    bool isInit = false;
    std::vector<int> lm_IDs;  // of instance 0 (filter.h:70)
    std::vector<float> map;   // true map as [id, x, y] triplets, for localisation-only filters (filter.h:69)
    virtual ~Filter() = default;
    // trueMapCallback stores `filter->map = msg->data` (localization_node.cpp:152-156); a batched filter also uploads it
    virtual void setTrueMap(const std::vector<float>& id_x_y) { map = id_x_y; }
    // Filter::setupStatePublisher(ros::NodeHandle) (filter.h:65) advertises the state topic that publishState() feeds every
    // tick (localization_node.cpp:139).  Without ROS there is nothing to advertise (publishState() leaves the payload in
    // `last_state` of the concrete class), but the batched EKF / UKF use the call for what it announces: instance 0's state
    // will be read after every update, so they start tracking it (slam_track_instance) and the per-tick publishState does not
    // force one launch per tick for the whole batch.
    virtual void setupStatePublisher() {}
    virtual void readParams(const slam_config& config) = 0;                     // filter.h:59 (YAML::Node there)
    virtual void init(float x_0, float y_0, float yaw_0) = 0;                   // filter.h:60
    virtual void update(Command::ConstPtr cmdMsg, Float32MultiArray::ConstPtr lmMeasMsg) = 0;  // filter.h:61
    virtual void publishState() = 0;                                            // filter.h:66
    virtual std::vector<double> getStateVector() { throw std::runtime_error("getStateVector is not defined for this filter."); }  // filter.h:76
    // filter.h:74: only the pose graph overrides it
    virtual void updateNaiveVehPoseEstimate(const std::vector<double>&, const std::vector<int>&) { throw std::runtime_error("updateNaiveVehPoseEstimate is not defined for this filter."); }
    FilterChoice filter_to_compare = FilterChoice::NOT_SET;   // filter.h:72
};

// Batch of B EKF-SLAM instances behind the single-instance interface.  update(cmd, meas) applies the SAME message
// to every instance (a drop-in for one filter when B == 1); updateBatch / updateSim are the batched entry points.
class BatchedEKF : public Filter {
public:
    BatchedEKF(int batch, int L_max, int device = 0) : batch_(batch), L_max_(L_max), device_(device) {
        type = FilterChoice::EKF_SLAM;
        check(slam_config_default(&cfg_));
    }
    ~BatchedEKF() override { if (h_) slam_destroy(h_); }
    BatchedEKF(const BatchedEKF&) = delete;
    BatchedEKF& operator=(const BatchedEKF&) = delete;

    void readParams(const slam_config& config) override {
        cfg_ = config;
        if (h_) { slam_destroy(h_); h_ = nullptr; }
        check(slam_create(&cfg_, SLAM_EKF_SLAM, batch_, L_max_, SLAM_F64, device_, &h_));
    }
    void readParamsFile(const std::string& params_yaml) {  // localization_node.cpp:29-30
        slam_config c;
        check(slam_config_default(&c));
        check(slam_config_load(&c, params_yaml.c_str()));
        readParams(c);
    }
    void setupStatePublisher() override { trackInstance(0); }
    void trackInstance(int instance) { need(); check(slam_track_instance(h_, instance)); }   // -1 = off
    void init(float x_0, float y_0, float yaw_0) override {
        need();
        check(slam_init(h_, x_0, y_0, yaw_0));
        isInit = true;
    }
    void update(Command::ConstPtr cmdMsg, Float32MultiArray::ConstPtr lmMeasMsg) override {
        need();
        const int k = (int)(lmMeasMsg->data.size() / 3);   // ekf.cpp:65
        const int ks = k > 0 ? k : 1;
        std::vector<float> meas((size_t)batch_ * ks * 3, 0.f);
        std::vector<int32_t> cnt(batch_, k);
        for (int b = 0; b < batch_; ++b)
            for (int i = 0; i < 3 * k; ++i) meas[(size_t)b * ks * 3 + i] = lmMeasMsg->data[i];
        const float cmd[2] = {cmdMsg->fwd, cmdMsg->ang};
        check(slam_step(h_, cmd, meas.data(), cnt.data(), ks));
    }
    void updateBatch(const Command& cmd, const float* meas, const int32_t* meas_count, int k_stride) {
        need();
        const float c[2] = {cmd.fwd, cmd.ang};
        check(slam_step(h_, c, meas, meas_count, k_stride));
    }
    void setMap(const std::vector<double>& map_xy) { need(); check(slam_set_map(h_, map_xy.data(), (int)(map_xy.size() / 2))); }
    void setSeed(uint64_t seed) { need(); check(slam_set_seed(h_, seed)); }
    void updateSim(const Command& cmd) {  // device-side get_cmd (sim_node.py:209-250) + update
        need();
        const float c[2] = {cmd.fwd, cmd.ang};
        check(slam_step_sim(h_, c));
    }
    std::vector<double> getStateVector() override { return getStateVector(0); }   // ekf.cpp:181-184
    std::vector<double> getStateVector(int instance) {
        need();
        std::vector<double> x(slam_state_dim_max(h_));
        int32_t M = 0;
        std::vector<int32_t> ids(L_max_);
        check(slam_get_state(h_, instance, x.data(), nullptr, &M, ids.data(), nullptr));
        x.resize(3 + 2 * M);
        if (instance == 0) lm_IDs.assign(ids.begin(), ids.begin() + M);
        return x;
    }
    // EKF::publishState (ekf.cpp:192-220): build the message payload; `last_state` is what would be published.
    void publishState() override { last_state = stateMsg(0); }
    EKFState stateMsg(int instance) {
        need();
        const int nmax = slam_state_dim_max(h_);
        std::vector<double> x(nmax), P((size_t)nmax * nmax);
        std::vector<int32_t> ids(L_max_);
        EKFState s;
        check(slam_get_state(h_, instance, x.data(), P.data(), &s.M, ids.data(), &s.timestep));
        const int n = 3 + 2 * s.M;
        s.x_v = (float)x[0]; s.y_v = (float)x[1]; s.yaw_v = (float)x[2];
        for (int i = 0; i < s.M; ++i) { s.landmarks.push_back((float)ids[i]); s.landmarks.push_back((float)x[3 + 2 * i]); s.landmarks.push_back((float)x[4 + 2 * i]); }
        s.P.resize((size_t)n * n);
        for (size_t i = 0; i < (size_t)n * n; ++i) s.P[i] = (float)P[i];
        return s;
    }
    std::vector<double> errorStats() { need(); std::vector<double> e(batch_); check(slam_error_stats(h_, e.data())); return e; }
    slam_handle* handle() { return h_; }
    EKFState last_state;

private:
    void need() const { if (!h_) throw std::runtime_error("readParams() has not been called"); }
    slam_config cfg_{};
    slam_handle* h_ = nullptr;
    int batch_, L_max_, device_;
};

// UKFState.msg payload as UKF::publishState fills it (ukf.cpp:60-104)
struct UKFState {
    int32_t timestep = 0;
    float x_v = 0, y_v = 0, yaw_v = 0;
    int32_t M = 0;
    std::vector<float> landmarks;  // [id, x, y] * M
    std::vector<float> P;          // row-major (4+2M)^2
    std::vector<float> X;          // sigma points of the last prediction stage, column by column (ukf.cpp:92-101)
};

// Batch of B UKF-SLAM instances behind the reference's UKF interface (filter.h:177-223, ukf.cpp).
class BatchedUKF : public Filter {
public:
    BatchedUKF(int batch, int L_max, int device = 0) : batch_(batch), L_max_(L_max), device_(device) {
        type = FilterChoice::UKF_SLAM;
        check(slam_config_default(&cfg_));
    }
    ~BatchedUKF() override { if (h_) slam_destroy(h_); }
    BatchedUKF(const BatchedUKF&) = delete;
    BatchedUKF& operator=(const BatchedUKF&) = delete;
    void readParams(const slam_config& config) override {
        cfg_ = config;
        if (h_) { slam_destroy(h_); h_ = nullptr; }
        check(slam_create(&cfg_, kind_, batch_, L_max_, SLAM_F64, device_, &h_));
    }
    void init(float x_0, float y_0, float yaw_0) override { need(); check(slam_init(h_, x_0, y_0, yaw_0)); isInit = true; }
    void update(Command::ConstPtr cmdMsg, Float32MultiArray::ConstPtr lmMeasMsg) override {   // ukf.cpp:161-195
        need();
        const int k = (int)(lmMeasMsg->data.size() / 3);
        const int ks = k > 0 ? k : 1;
        std::vector<float> meas((size_t)batch_ * ks * 3, 0.f);
        std::vector<int32_t> cnt(batch_, k);
        for (int b = 0; b < batch_; ++b)
            for (int i = 0; i < 3 * k; ++i) meas[(size_t)b * ks * 3 + i] = lmMeasMsg->data[i];
        const float cmd[2] = {cmdMsg->fwd, cmdMsg->ang};
        check(slam_step(h_, cmd, meas.data(), cnt.data(), ks));
    }
    // the two public halves of UKF::update (filter.h:187-188); measurements as DEVICE pointers
    void predictionStage(const Command& cmd) { need(); const float c[2] = {cmd.fwd, cmd.ang}; check(slam_predict(h_, c)); }
    void updateStage(const float* d_meas, const int32_t* d_meas_count, int k_stride) { need(); check(slam_update_dev(h_, d_meas, d_meas_count, k_stride)); }
    void setMap(const std::vector<double>& map_xy) { need(); check(slam_set_map(h_, map_xy.data(), (int)(map_xy.size() / 2))); }
    void updateSim(const Command& cmd) { need(); const float c[2] = {cmd.fwd, cmd.ang}; check(slam_step_sim(h_, c)); }
    // ukf.cpp:47-53 (x, y, yaw, landmarks...); the reference's fixed-size Vector3d bug is not replicated
    std::vector<double> getStateVector() override {
        need();
        std::vector<double> x(slam_state_dim_max(h_));
        int32_t M = 0;
        std::vector<int32_t> ids(L_max_);
        check(slam_get_state(h_, 0, x.data(), nullptr, &M, ids.data(), nullptr));
        lm_IDs.assign(ids.begin(), ids.begin() + M);
        std::vector<double> out = {x[0], x[1], std::remainder(std::atan2(x[3], x[2]), 2 * 3.14159265358979323846)};
        out.insert(out.end(), x.begin() + 4, x.begin() + 4 + 2 * M);
        return out;
    }
    void publishState() override { last_state = stateMsg(0); }
    UKFState stateMsg(int instance) {
        need();
        const int nmax = slam_state_dim_max(h_);
        std::vector<double> x(nmax), P((size_t)nmax * nmax), X((size_t)nmax * (2 * nmax + 1));
        std::vector<int32_t> ids(L_max_);
        UKFState s;
        check(slam_get_state(h_, instance, x.data(), P.data(), &s.M, ids.data(), &s.timestep));
        const int n = 4 + 2 * s.M;
        s.x_v = (float)x[0]; s.y_v = (float)x[1]; s.yaw_v = (float)std::remainder(std::atan2(x[3], x[2]), 2 * 3.14159265358979323846);
        for (int i = 0; i < s.M; ++i) { s.landmarks.push_back((float)ids[i]); s.landmarks.push_back((float)x[4 + 2 * i]); s.landmarks.push_back((float)x[5 + 2 * i]); }
        s.P.resize((size_t)n * n);
        for (size_t i = 0; i < (size_t)n * n; ++i) s.P[i] = (float)P[i];
        int32_t rows = 0, cols = 0;
        check(slam_get_sigma_points(h_, instance, X.data(), &rows, &cols));
        s.X.resize((size_t)rows * cols);
        for (size_t i = 0; i < (size_t)rows * cols; ++i) s.X[i] = (float)X[i];
        return s;
    }
    std::vector<double> errorStats() { need(); std::vector<double> e(batch_); check(slam_error_stats(h_, e.data())); return e; }
    slam_handle* handle() { return h_; }
    UKFState last_state;

protected:
    int kind_ = SLAM_UKF_SLAM;

private:
    void need() const { if (!h_) throw std::runtime_error("readParams() has not been called"); }
    slam_config cfg_{};
    slam_handle* h_ = nullptr;
    int batch_, L_max_, device_;
};

// NaiveFilter (filter.h:325-369): propagate the commands, ignore the measurements.  The pose graph's default secondary
// filter (params.yaml:60).
class NaiveFilter : public Filter {
public:
    NaiveFilter() { type = FilterChoice::NAIVE_COMMAND_PROPAGATION; }
    void readParams(const slam_config&) override {}
    void init(float x_0, float y_0, float yaw_0) override { timestep = 0; x_t = {x_0, y_0, yaw_0}; isInit = true; }
    void update(Command::ConstPtr cmdMsg, Float32MultiArray::ConstPtr) override {
        timestep += 1;
        const double x = x_t[0] + (double)cmdMsg->fwd * std::cos(x_t[2]), y = x_t[1] + (double)cmdMsg->fwd * std::sin(x_t[2]);
        x_t = {x, y, std::remainder(x_t[2] + (double)cmdMsg->ang, 2 * 3.14159265358979323846)};
    }
    void publishState() override {}
    std::vector<double> getStateVector() override { return x_t; }
    int timestep = 0;
    std::vector<double> x_t{0, 0, 0};
};

// PoseGraphState.msg payload as PoseGraph::publishState fills it (pose_graph.cpp:302-387)
struct PoseGraphState {
    int32_t timestep = 0, M = 0;
    std::vector<float> x_v, y_v, yaw_v;        // poses 0 .. timestep-1 (the reference's loop is i < timestep)
    std::vector<float> landmarks;              // [x, y] * M
    std::vector<int32_t> meas_connections;     // [pose, landmark index (-1: first detection)] * k
    bool is_result = false;                    // /state/pose_graph/result vs /state/pose_graph/initial
};

// Batch of B pose graphs behind the reference's PoseGraph interface (filter.h:232-322, pose_graph.cpp, GTSAM path).
// update(cmd, meas) applies the SAME message to every instance; updateBatch takes per-instance measurements.
class BatchedPoseGraph : public Filter {
public:
    BatchedPoseGraph(int batch, int num_iterations, int L_max, int k_per_pose = 8, int device = 0)
        : num_iterations_total(num_iterations), batch_(batch), L_max_(L_max), kp_(k_per_pose), device_(device) {
        type = FilterChoice::POSE_GRAPH_SLAM;
        filter_to_compare = FilterChoice::NAIVE_COMMAND_PROPAGATION;   // params.yaml:60
        check(slam_config_default(&cfg_));
    }
    ~BatchedPoseGraph() override { if (h_) pgs_destroy(h_); }
    BatchedPoseGraph(const BatchedPoseGraph&) = delete;
    BatchedPoseGraph& operator=(const BatchedPoseGraph&) = delete;
    bool solve_graph_every_iteration = false;   // params.yaml:64
    bool solved_pose_graph = false;
    int timestep = 0;
    int num_iterations_total;

    void readParams(const slam_config& config) override {   // pose_graph.cpp:12-66
        cfg_ = config;
        if (h_) { pgs_destroy(h_); h_ = nullptr; }
        check(pgs_create(&cfg_, batch_, num_iterations_total, L_max_, kp_, device_, &h_));
    }
    void init(float x_0, float y_0, float yaw_0) override {  // pose_graph.cpp:68-95
        need();
        check(pgs_init(h_, x_0, y_0, yaw_0));
        isInit = true; solved_pose_graph = false; timestep = 0; have_sec_ = false;
    }
    // pose_graph.cpp:97-119: the same secondary estimate for every instance (e.g. the NaiveFilter)
    void updateNaiveVehPoseEstimate(const std::vector<double>& state_vector, const std::vector<int>&) override {
        sec_.resize((size_t)batch_ * 3);
        for (int b = 0; b < batch_; ++b) for (int c = 0; c < 3; ++c) sec_[(size_t)3 * b + c] = state_vector[c];
        have_sec_ = true;
    }
    void update(Command::ConstPtr cmdMsg, Float32MultiArray::ConstPtr lmMeasMsg) override {   // pose_graph.cpp:199-267
        const int k = (int)(lmMeasMsg->data.size() / 3);
        std::vector<float> meas((size_t)batch_ * (k > 0 ? k : 1) * 3, 0.f);
        std::vector<int32_t> cnt(batch_, k);
        for (int b = 0; b < batch_; ++b) for (int i = 0; i < 3 * k; ++i) meas[(size_t)b * k * 3 + i] = lmMeasMsg->data[i];
        updateBatch(*cmdMsg, meas.data(), cnt.data(), k);
    }
    void updateBatch(const Command& cmd, const float* meas, const int32_t* meas_count, int k_stride) {
        need();
        if (solved_pose_graph && !solve_graph_every_iteration) return;          // :201-205
        if (timestep + 1 >= num_iterations_total) { solvePoseGraph(); publishState(); return; }   // :208-214
        const float c[2] = {cmd.fwd, cmd.ang};
        check(pgs_update(h_, c, k_stride > 0 ? meas : nullptr, k_stride > 0 ? meas_count : nullptr, k_stride, have_sec_ ? sec_.data() : nullptr));
        timestep += 1;
        if (solve_graph_every_iteration) { solvePoseGraph(); check(pgs_adopt_result(h_)); }   // :258-264
        publishState();
    }
    void solvePoseGraph() { need(); check(pgs_solve(h_)); solved_pose_graph = true; }    // pose_graph.cpp:269-300
    void publishState() override { last_state = stateMsg(0); }
    PoseGraphState stateMsg(int instance) {
        need();
        PoseGraphState s;
        std::vector<double> poses((size_t)3 * (timestep + 1)), lms((size_t)2 * L_max_);
        std::vector<int32_t> ids(L_max_);
        check(pgs_get_graph(h_, instance, solved_pose_graph ? 1 : 0, poses.data(), lms.data(), &s.timestep, &s.M, ids.data()));
        for (int i = 0; i < s.timestep; ++i) { s.x_v.push_back((float)poses[3 * i]); s.y_v.push_back((float)poses[3 * i + 1]); s.yaw_v.push_back((float)poses[3 * i + 2]); }
        for (int i = 0; i < 2 * s.M; ++i) s.landmarks.push_back((float)lms[i]);
        int32_t n = 0;
        s.meas_connections.resize((size_t)2 * (timestep + 1) * kp_);
        check(pgs_get_connections(h_, instance, s.meas_connections.data(), (timestep + 1) * kp_, &n));
        s.meas_connections.resize((size_t)2 * n);
        s.is_result = solved_pose_graph;
        if (instance == 0) lm_IDs.assign(ids.begin(), ids.begin() + s.M);
        return s;
    }
    pgs_handle* handle() { return h_; }
    PoseGraphState last_state;

private:
    void need() const { if (!h_) throw std::runtime_error("readParams() has not been called"); }
    slam_config cfg_{};
    pgs_handle* h_ = nullptr;
    int batch_, L_max_, kp_, device_;
    std::vector<double> sec_;
    bool have_sec_ = false;
};

// UKF localisation against the known map (FilterChoice::UKF_LOC, localization_node.cpp:39-41, ukf.cpp:146-154): the state is
// the vehicle only; every detection updates against the true-map landmark of the same id.
class BatchedUKFLoc : public BatchedUKF {
public:
    explicit BatchedUKFLoc(int batch, int device = 0) : BatchedUKF(batch, 1, device) { type = FilterChoice::UKF_LOC; kind_ = SLAM_UKF_LOC; }
    void setTrueMap(const std::vector<float>& id_x_y) override {   // rows are sorted by id in the reference's message (sim_node.py:203)
        map = id_x_y;
        std::vector<double> xy(2 * (id_x_y.size() / 3));
        for (size_t i = 0; i < id_x_y.size() / 3; ++i) { xy[2 * i] = id_x_y[3 * i + 1]; xy[2 * i + 1] = id_x_y[3 * i + 2]; }
        setMap(xy);
    }
};

// ROS-free restatement of the node harness (localization_node.cpp): the two FIFO queues the subscriber callbacks fill
// (:17-18, :142-150), the true-map gate for localisation-only filters (:21, :113-116, :152-156), the secondary filter hook
// for the pose graph (:24-25, :123-128) and iterate() (:108-140), which a ROS timer calls at 1/dt.  One (command, measurement)
// pair is consumed per call; a call that finds a queue empty, the filter uninitialised or the map missing returns early.
struct LocalizationNode {
    std::unique_ptr<Filter> filter, filter_secondary;        // :24-25
    std::queue<Command::ConstPtr> cmdQueue;                  // :17
    std::queue<Float32MultiArray::ConstPtr> lmMeasQueue;     // :18
    bool loadedTrueMap = false;                              // :21
    int iterations = 0;

    void initCallback(float x_0, float y_0, float yaw_0) {   // :90-106
        if (filter->isInit) return;
        filter->init(x_0, y_0, yaw_0);
        if (filter->filter_to_compare != FilterChoice::NOT_SET) filter_secondary->init(x_0, y_0, yaw_0);
    }
    void cmdCallback(const Command::ConstPtr& msg) { cmdQueue.push(msg); }                    // :142-145
    void lmMeasCallback(const Float32MultiArray::ConstPtr& msg) { lmMeasQueue.push(msg); }    // :147-150
    void trueMapCallback(const Float32MultiArray::ConstPtr& msg) { filter->setTrueMap(msg->data); loadedTrueMap = true; }   // :152-156
    bool iterate() {                                         // :108-140
        if (!filter->isInit || cmdQueue.empty() || lmMeasQueue.empty()) return false;        // :109-112
        if (filter->type == FilterChoice::UKF_LOC && !loadedTrueMap) return false;           // :113-116
        Command::ConstPtr cmdMsg = cmdQueue.front(); cmdQueue.pop();                         // :118-119
        Float32MultiArray::ConstPtr lmMeasMsg = lmMeasQueue.front(); lmMeasQueue.pop();      // :120-121
        if (filter->filter_to_compare != FilterChoice::NOT_SET) {                            // :124-128
            filter_secondary->update(cmdMsg, lmMeasMsg);
            filter->updateNaiveVehPoseEstimate(filter_secondary->getStateVector(), filter_secondary->lm_IDs);
        }
        filter->update(cmdMsg, lmMeasMsg);                                                   // :131
        if (filter->filter_to_compare != FilterChoice::NOT_SET) filter_secondary->publishState();   // :135-139
        else filter->publishState();
        iterations += 1;
        return true;
    }
};

}  // namespace slam_amd
