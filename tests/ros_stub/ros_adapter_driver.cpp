// ros_adapter_driver.cpp — drives include/slam_filter_ros.hpp the way localization_node.cpp drives a filter: factory -> readParams(config)
// -> setupStatePublisher(node) -> init -> per tick update(cmdMsg, lmMeasMsg) + publishState(), all through std::unique_ptr<Filter>.
// usage: ros_adapter_driver <ekf|ukf> <batch> <L_max> <stream.txt> <dump.bin> [id_known]
// stream: one line per tick "fwd ang k {id range bearing}*k"; dump: per tick the published message of instance 0 as float32
// [timestep, x_v, y_v, yaw_v, M, landmarks(3M), P(n*n)] preceded by its length (int32), then getStateVector() of the last tick (float64,
// preceded by MINUS its length).
#include "ros_stub.hpp"
#define SLAM_AMD_USE_ROS_MSGS 1
#include "../../include/slam_filter_ros.hpp"

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>

int main(int argc, char** argv) {
    if (argc < 6) { std::fprintf(stderr, "usage: ros_adapter_driver <ekf|ukf> <batch> <L_max> <stream.txt> <dump.bin> [id_known]\n"); return 2; }
    try {
        const std::string kind = argv[1];
        const int B = std::atoi(argv[2]), L = std::atoi(argv[3]);
        std::unique_ptr<Filter> filter;                                   // localization_node.cpp:26
        if (kind == "ekf") filter = std::make_unique<slam_amd::BatchedEKFRos>(B, L);
        else filter = std::make_unique<slam_amd::BatchedUKFRos>(B, L);   // the ONE added factory branch
        YAML::Node config;                                                // what YAML::LoadFile(params.yaml) would hold
        config.child("constraints").child("measurements").child("landmark_id_is_known").set(argc > 6 ? std::atof(argv[6]) : 1.0);
        config.child("constraints").child("measurements").child("min_landmark_separation").set(0.1);
        filter->readParams(config);                                       // localization_node.cpp:47
        ros::NodeHandle node;
        filter->setupStatePublisher(node);                                // localization_node.cpp:187
        filter->init(0.f, 0.f, 0.f);                                      // initCallback, localization_node.cpp:100
        std::ifstream in(argv[4]);
        if (!in) throw std::runtime_error("cannot open the stream");
        std::ofstream out(argv[5], std::ios::binary);
        std::string line;
        int ticks = 0;
        while (std::getline(in, line)) {
            if (line.empty()) continue;
            std::istringstream ls(line);
            auto cmd = std::make_shared<base_pkg::Command>();
            auto lm = std::make_shared<std_msgs::Float32MultiArray>();
            int k = 0;
            ls >> cmd->fwd >> cmd->ang >> k;
            lm->data.resize((size_t)3 * k);
            for (auto& v : lm->data) ls >> v;
            filter->update(cmd, lm);                                      // localization_node.cpp:131
            filter->publishState();                                       // localization_node.cpp:138
            std::vector<float> rec;
            if (kind == "ekf") {
                const base_pkg::EKFState& s = filter->statePub.last<base_pkg::EKFState>();
                rec = {(float)s.timestep, s.x_v, s.y_v, s.yaw_v, (float)s.M};
                rec.insert(rec.end(), s.landmarks.begin(), s.landmarks.end());
                rec.insert(rec.end(), s.P.begin(), s.P.end());
            } else {
                const base_pkg::UKFState& s = filter->statePub.last<base_pkg::UKFState>();
                rec = {(float)s.timestep, s.x_v, s.y_v, s.yaw_v, (float)s.M};
                rec.insert(rec.end(), s.landmarks.begin(), s.landmarks.end());
                rec.insert(rec.end(), s.P.begin(), s.P.end());
            }
            const int32_t len = (int32_t)rec.size();
            out.write((const char*)&len, sizeof(len));
            out.write((const char*)rec.data(), sizeof(float) * rec.size());
            ticks += 1;
        }
        const Eigen::VectorXd sv = filter->getStateVector();              // localization_node.cpp:127 (as a secondary filter)
        const int32_t n = (int32_t)sv.size(), tag = -n;                  // (negative length: the float64 record)
        out.write((const char*)&tag, sizeof(tag));
        for (int i = 0; i < n; ++i) { const double v = sv(i); out.write((const char*)&v, sizeof(v)); }
        std::printf("adapter ok: kind=%s batch=%d ticks=%d topic=%s lm_IDs=%zu\n", kind.c_str(), B, ticks, filter->statePub.topic.c_str(), filter->lm_IDs.size());
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "adapter error: %s\n", e.what());
        return 3;
    }
}
