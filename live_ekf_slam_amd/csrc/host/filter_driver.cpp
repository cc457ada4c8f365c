// filter_driver.cpp — ROS-free restatement of the node harness loop (localization_node.cpp:108-140 `iterate`):
// FIFO-paired (command, measurement) messages are popped one pair per tick and fed to Filter::update, then
// publishState().  Here the filter is the batched MI355X engine behind the reference's Filter interface.
//
// usage: filter_driver <batch> <L> <steps>   (synthetic straight-ish trajectory, device-side measurements)
// Prints one line: mean per-instance average position error and the published state size of instance 0.
#include <cstdio>
#include <cstdlib>
#include <queue>
#include <random>

#include "../../../include/slam_filter.hpp"

using namespace slam_amd;

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, L = argc > 2 ? atoi(argv[2]) : 20, T = argc > 3 ? atoi(argv[3]) : 100;
    try {
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
