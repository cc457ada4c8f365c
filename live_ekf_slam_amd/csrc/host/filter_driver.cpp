// filter_driver.cpp — ROS-free restatement of the node harness loop (localization_node.cpp:108-140 `iterate`):
// FIFO-paired (command, measurement) messages are popped one pair per tick and fed to Filter::update, then
// publishState().  Here the filter is the batched MI355X engine behind the reference's Filter interface.
//
// usage: filter_driver <batch> <L> <steps> [pose_graph | ukf]   (synthetic straight-ish trajectory, device-side measurements)
// Prints one line: mean per-instance average position error and the published state size of instance 0.
#include <cstdio>
#include <cstdlib>
#include <queue>
#include <random>

#include "../../../include/slam_filter.hpp"

using namespace slam_amd;

// `filter: "pose_graph"` (params.yaml:11): the node runs a secondary filter first and hands its estimate to the pose
// graph every tick (localization_node.cpp:124-131); the pose graph solves when timestep+1 >= num_iterations.
static int run_pose_graph(int B, int L, int T) {
    std::unique_ptr<Filter> filter = std::make_unique<BatchedPoseGraph>(B, /*num_iterations=*/T, L);   // localization_node.cpp:45-46
    std::unique_ptr<Filter> filter_secondary = std::make_unique<NaiveFilter>();                          // :68-69
    auto* pg = static_cast<BatchedPoseGraph*>(filter.get());
    slam_config cfg;
    check(slam_config_default(&cfg));
    filter->readParams(cfg); filter_secondary->readParams(cfg);
    filter->init(0.f, 0.f, 0.f); filter_secondary->init(0.f, 0.f, 0.f);                                 // :100-105
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<float> U(-0.02f, 0.02f);
    for (int t = 0; t < T; ++t) {                                                                         // iterate :108-140
        auto cmd = std::make_shared<Command>(); cmd->fwd = 0.1f; cmd->ang = (t / 40) % 2 ? -0.03f : 0.03f;
        auto meas = std::make_shared<Float32MultiArray>();
        if (t % 3 == 0) meas->data = {(float)(t / 30), 2.0f + U(rng), 0.3f + U(rng)};                    // a landmark seen now and then
        filter_secondary->update(cmd, meas);                                                              // :125
        filter->updateNaiveVehPoseEstimate(filter_secondary->getStateVector(), filter_secondary->lm_IDs); // :127
        filter->update(cmd, meas);                                                                        // :131
    }
    std::printf("driver ok: pose_graph batch=%d poses=%d solved=%d M0=%d result_topic=%d x_len=%zu conns=%zu\n", B, pg->timestep + 1,
                (int)pg->solved_pose_graph, pg->last_state.M, (int)pg->last_state.is_result, pg->last_state.x_v.size(),
                pg->last_state.meas_connections.size() / 2);
    return 0;
}

// `filter: "ukf_slam"` (localization_node.cpp:36-38): the same iterate() loop with the UKF behind the Filter pointer
static int run_ukf(int B, int L, int T) {
    std::unique_ptr<Filter> filter = std::make_unique<BatchedUKF>(B, L);
    auto* ukf = static_cast<BatchedUKF*>(filter.get());
    slam_config cfg;
    check(slam_config_default(&cfg));
    filter->readParams(cfg);
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-10.0, 10.0);
    std::vector<double> map(2 * L);
    for (auto& v : map) v = U(rng);
    ukf->setMap(map);
    filter->init(0.f, 0.f, 0.f);
    for (int t = 0; t < T; ++t) {
        Command c; c.fwd = 0.1f; c.ang = (t / 40) % 2 ? -0.03f : 0.03f;
        ukf->updateSim(c);
        filter->publishState();
    }
    auto cmd = std::make_shared<Command>(); cmd->fwd = 0.05f; cmd->ang = 0.01f;
    filter->update(cmd, std::make_shared<Float32MultiArray>());
    filter->publishState();
    const std::vector<double> sv = filter->getStateVector();
    double mean = 0;
    for (double e : ukf->errorStats()) mean += e;
    std::printf("driver ok: ukf batch=%d L=%d steps=%d mean_avg_err=%.6f M0=%d timestep=%d P_len=%zu X_len=%zu sv_len=%zu\n", B, L, T + 1,
                mean / B, ukf->last_state.M, ukf->last_state.timestep, ukf->last_state.P.size(), ukf->last_state.X.size(), sv.size());
    return 0;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, L = argc > 2 ? atoi(argv[2]) : 20, T = argc > 3 ? atoi(argv[3]) : 100;
    try {
        if (argc > 4 && std::string(argv[4]) == "pose_graph") return run_pose_graph(B, L, T);
        if (argc > 4 && std::string(argv[4]) == "ukf") return run_ukf(B, L, T);
        std::unique_ptr<Filter> filter = std::make_unique<BatchedEKF>(B, L);   // localization_node.cpp:33-35
        auto* ekf = static_cast<BatchedEKF*>(filter.get());
        slam_config cfg;
        check(slam_config_default(&cfg));
        filter->readParams(cfg);                                                // localization_node.cpp:47
        std::mt19937_64 rng(7);
        std::uniform_real_distribution<double> U(-10.0, 10.0);
        std::vector<double> map(2 * L);
        for (auto& v : map) v = U(rng);
        ekf->setMap(map);
        filter->init(0.f, 0.f, 0.f);                                            // initCallback :100
        std::queue<Command> cmdQueue;                                           // localization_node.cpp:17
        for (int t = 0; t < T; ++t) { Command c; c.fwd = 0.1f; c.ang = (t / 40) % 2 ? -0.03f : 0.03f; cmdQueue.push(c); }
        while (filter->isInit && !cmdQueue.empty()) {                           // iterate :109-121
            const Command c = cmdQueue.front();
            cmdQueue.pop();
            ekf->updateSim(c);                                                  // :131 (measurements generated on the device)
            filter->publishState();                                             // :135-139
        }
        // the single-message path of the reference interface: same (cmd, meas) for every instance
        auto cmd = std::make_shared<Command>(); cmd->fwd = 0.05f; cmd->ang = 0.01f;
        auto meas = std::make_shared<Float32MultiArray>();
        filter->update(cmd, meas);                                              // empty detection list (ekf.cpp:67-71)
        filter->publishState();
        double mean = 0;
        for (double e : ekf->errorStats()) mean += e;
        std::printf("driver ok: batch=%d L=%d steps=%d mean_avg_err=%.6f M0=%d timestep=%d P_len=%zu\n", B, L, T + 1,
                    mean / B, ekf->last_state.M, ekf->last_state.timestep, ekf->last_state.P.size());
    } catch (const std::exception& e) {
        std::fprintf(stderr, "driver failed: %s\n", e.what());
        return 1;
    }
    return 0;
}
