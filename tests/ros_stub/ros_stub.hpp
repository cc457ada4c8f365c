// ros_stub.hpp — TEST stand-ins for the handful of ROS / yaml-cpp / Eigen names include/slam_filter_ros.hpp touches, so that the adapter
// can be compiled, linked and RUN in an image that has none of those packages (tests/test_ros_adapter_*.py).  Not a substitute for
// building inside the reference's localization_pkg: only the members the adapter uses exist, written from the interface the adapter
// relies on (filter.h:44-77 for the abstract class; the .msg field lists for the messages).  Test infrastructure only.
#pragma once
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace ros {
struct Publisher {
    std::shared_ptr<std::vector<std::shared_ptr<void>>> sent = std::make_shared<std::vector<std::shared_ptr<void>>>();
    std::string topic;
    template <class M> void publish(const M& m) { sent->push_back(std::make_shared<M>(m)); }
    template <class M> const M& last() const { return *std::static_pointer_cast<M>(sent->back()); }
};
struct NodeHandle {
    template <class M> Publisher advertise(const std::string& topic, int) { Publisher p; p.topic = topic; return p; }
};
}  // namespace ros

namespace YAML {
// a tree of string-keyed nodes with scalar leaves: what config["a"]["b"].as<T>() needs
class Node {
public:
    Node() = default;
    explicit Node(double v) : scalar_(v), has_(true) {}
    Node operator[](const std::string& k) const {
        auto it = kids_.find(k);
        return it == kids_.end() ? Node() : *it->second;
    }
    Node& child(const std::string& k) {
        auto& p = kids_[k];
        if (!p) p = std::make_shared<Node>();
        defined_ = true;
        return *p;
    }
    void set(double v) { scalar_ = v; has_ = true; defined_ = true; }
    explicit operator bool() const { return has_ || defined_; }
    template <class T> T as() const {
        if (!has_) throw std::runtime_error("bad conversion");
        return static_cast<T>(scalar_);
    }
private:
    std::map<std::string, std::shared_ptr<Node>> kids_;
    double scalar_ = 0.0;
    bool has_ = false, defined_ = false;
};
template <> inline bool Node::as<bool>() const {
    if (!has_) throw std::runtime_error("bad conversion");
    return scalar_ != 0.0;
}
}  // namespace YAML

namespace Eigen {
class VectorXd {
public:
    VectorXd() = default;
    explicit VectorXd(long n) : v_((size_t)n, 0.0) {}
    double& operator()(long i) { return v_[(size_t)i]; }
    double operator()(long i) const { return v_[(size_t)i]; }
    long size() const { return (long)v_.size(); }
private:
    std::vector<double> v_;
};
}  // namespace Eigen

namespace std_msgs {
struct Float32MultiArray {
    std::vector<float> data;
    typedef std::shared_ptr<const Float32MultiArray> ConstPtr;
};
}  // namespace std_msgs

namespace base_pkg {
struct Command {
    float fwd = 0.f, ang = 0.f;
    typedef std::shared_ptr<const Command> ConstPtr;
};
struct EKFState {
    int timestep = 0;
    float x_v = 0.f, y_v = 0.f, yaw_v = 0.f;
    int M = 0;
    std::vector<float> landmarks, P;
};
struct UKFState {
    int timestep = 0;
    float x_v = 0.f, y_v = 0.f, yaw_v = 0.f;
    int M = 0;
    std::vector<float> landmarks, P, X, X_pred;
};
}  // namespace base_pkg

enum class FilterChoice { NOT_SET = 0, EKF_SLAM, UKF_LOC, UKF_SLAM, POSE_GRAPH_SLAM, NAIVE_COMMAND_PROPAGATION };

// the abstract class the node holds in std::unique_ptr<Filter> (filter.h:54-77): public members and virtuals only
class Filter {
public:
    FilterChoice type = FilterChoice::NOT_SET;
    Filter() {}
    virtual ~Filter() {}
    virtual void readParams(YAML::Node config) = 0;
    virtual void init(float x_0, float y_0, float yaw_0) = 0;
    virtual void update(base_pkg::Command::ConstPtr cmdMsg, std_msgs::Float32MultiArray::ConstPtr lmMeasMsg) = 0;
    ros::Publisher statePub;
    virtual void setupStatePublisher(ros::NodeHandle node) = 0;
    virtual void publishState() = 0;
    bool isInit = false;
    std::vector<float> map;
    std::vector<int> lm_IDs;
    FilterChoice filter_to_compare = FilterChoice::NOT_SET;
    virtual void updateNaiveVehPoseEstimate(Eigen::VectorXd, std::vector<int>) { throw std::runtime_error("updateNaiveVehPoseEstimate is not defined for this filter."); }
    virtual Eigen::VectorXd getStateVector() { throw std::runtime_error("getStateVector is not defined for this filter."); }
};
