// filter_driver.cpp — a ROS-free C++ host over the reference's Filter interface (include/slam_filter.hpp): the node harness
// of localization_node.cpp (both FIFO queues, the true-map gate, the secondary-filter hook, iterate()) driving the batched
// MI355X engine, with the scenario generators of sim_node.py in C++ (include/slam_scenario.hpp).  No Python anywhere.
//
//   filter_driver stream <ekf|ukf|ukf_loc> <batch> <L_max> <stream.txt> <dump.bin>
//       Feeds a recorded message stream (one line per tick: fwd ang k {id range bearing}*k; for ukf_loc a first line
//       "map L {id x y}*L") through cmdCallback / lmMeasCallback / trueMapCallback and iterate(), then dumps the state of
//       instances 0 and batch-1 (int64 M, n; then x[n], P[n*n] as doubles, each) for the parity test against the oracle.
//   filter_driver run <ekf|ukf> <batch> <L> <steps> [seed]
//       A BASELINE-style run: map + TSP commands from make_scenario(seed, L, steps) (reference generators, bit-exact), the
//       measurements generated per instance on the device (slam_run_sim / slam_step_sim); prints the error statistic.
//   filter_driver run_multi <ekf|ukf> <global batch> <L> <steps> <gpus> [seed] [gather: 0 host concat | 1 RCCL]
//       The same run with the global batch sharded over <gpus> devices of this node from ONE process (include/slam_multi.h):
//       contiguous shards, noise keyed by global instance id, no collective until the gather of the error statistics.
//   filter_driver pose_graph <batch> <L> <steps>
//       `filter: pose_graph` with the NaiveFilter secondary (localization_node.cpp:62-69,124-131) over the same harness.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "../../../include/slam_filter.hpp"
#include "../../../include/slam_multi.h"
#include "../../../include/slam_scenario.hpp"
#include "stream_parse.h"

using namespace slam_amd;

static std::unique_ptr<Filter> make_filter(const std::string& kind, int B, int L) {   // localization_node.cpp:33-47
    if (kind == "ekf") return std::make_unique<BatchedEKF>(B, L);
    if (kind == "ukf") return std::make_unique<BatchedUKF>(B, L);
    if (kind == "ukf_loc") return std::make_unique<BatchedUKFLoc>(B);
    throw std::runtime_error("Invalid filter choice: " + kind);                        // :44
}

static void dump_state(FILE* f, slam_handle* h, int inst, int base) {
    const int nmax = slam_state_dim_max(h);
    std::vector<double> x(nmax), P((size_t)nmax * nmax);
    int32_t M = 0;
    check(slam_get_state(h, inst, x.data(), P.data(), &M, nullptr, nullptr));
    const int64_t hdr[2] = {M, base + 2 * M};
    std::fwrite(hdr, sizeof(int64_t), 2, f);
    std::fwrite(x.data(), sizeof(double), (size_t)hdr[1], f);
    std::fwrite(P.data(), sizeof(double), (size_t)hdr[1] * hdr[1], f);
}

static int run_stream(const std::string& kind, int B, int L, const char* stream_path, const char* dump_path) {
    LocalizationNode node;
    node.filter = make_filter(kind, B, L);
    slam_config cfg;
    check(slam_config_default(&cfg));
    node.filter->readParams(cfg);                                                       // :47
    node.filter->setupStatePublisher();                                                 // main() :187 (nothing to advertise without ROS)
    std::ifstream in(stream_path);
    if (!in) throw std::runtime_error(std::string("cannot open ") + stream_path);
    std::string line;
    int ticks_without_input = 0;
    node.iterate();                                                                     // timer fires before anything arrived: early return
    node.initCallback(0.f, 0.f, 0.f);                                                   // :90-106
    int lineno = 0;
    while (std::getline(in, line)) {
        lineno += 1;
        if (line.find_first_not_of(" \t\r") == std::string::npos || line[line.find_first_not_of(" \t")] == '#') continue;
        slam_host::StreamLine sl;   // the file is untrusted text: bounded counts, finite values (host/stream_parse.h)
        std::string perr;
        if (!slam_host::parse_stream_line(line, &sl, &perr))
            throw std::runtime_error(std::string(stream_path) + ":" + std::to_string(lineno) + ": " + perr);
        if (sl.is_map) {                                                                // the simulator publishes the true map once
            auto m = std::make_shared<Float32MultiArray>();
            m->data = sl.data;
            if (kind == "ukf_loc" && node.iterate()) throw std::runtime_error("UKF_LOC iterated before the map arrived");
            node.trueMapCallback(m);
            continue;
        }
        auto cmd = std::make_shared<Command>();
        cmd->fwd = sl.fwd;
        cmd->ang = sl.ang;
        auto meas = std::make_shared<Float32MultiArray>();
        meas->data = sl.data;
        // the two topics arrive independently: the command first, a timer tick in between (iterate must wait for the
        // measurement, :109-112), then the measurement
        node.cmdCallback(cmd);
        {
            const bool must_wait = node.lmMeasQueue.empty();   // the command's measurement has not arrived yet
            const bool consumed = node.iterate();
            if (must_wait && consumed) throw std::runtime_error("iterate() consumed a command without its measurement");
            if (!consumed) ticks_without_input += 1;
        }
        node.lmMeasCallback(meas);
        if (node.cmdQueue.size() >= 3) while (node.iterate()) {}                        // a slow filter: the queues run ahead, then drain in FIFO order
    }
    while (node.iterate()) {}
    slam_handle* h = kind == "ekf" ? static_cast<BatchedEKF*>(node.filter.get())->handle() : static_cast<BatchedUKF*>(node.filter.get())->handle();
    FILE* f = std::fopen(dump_path, "wb");
    if (!f) throw std::runtime_error(std::string("cannot write ") + dump_path);
    dump_state(f, h, 0, kind == "ekf" ? 3 : 4);
    dump_state(f, h, B - 1, kind == "ekf" ? 3 : 4);
    std::fclose(f);
    std::printf("driver ok: stream kind=%s batch=%d iterations=%d early_returns=%d queues_left=%zu/%zu\n", kind.c_str(), B, node.iterations,
                ticks_without_input, node.cmdQueue.size(), node.lmMeasQueue.size());
    return 0;
}

static int run_scenario(const std::string& kind, int B, int L, int T, uint64_t seed) {
    const Scenario sc = make_scenario(seed, L, T);                                      // sim_node.py:155-206, 63-138
    std::unique_ptr<Filter> filter = make_filter(kind, B, L);
    slam_config cfg;
    check(slam_config_default(&cfg));
    filter->readParams(cfg);
    filter->init(0.f, 0.f, 0.f);
    std::vector<double> err;
    if (kind == "ekf") {
        auto* ekf = static_cast<BatchedEKF*>(filter.get());
        ekf->setMap(sc.map_xy);
        check(slam_run_sim(ekf->handle(), sc.cmds.data(), T));                           // every instance: get_cmd + EKF::update, T ticks
        filter->publishState();
        err = ekf->errorStats();
        std::printf("driver ok: run ekf batch=%d L=%d steps=%d M0=%d timestep=%d P_len=%zu ", B, L, T, ekf->last_state.M, ekf->last_state.timestep,
                    ekf->last_state.P.size());
    } else {
        auto* ukf = static_cast<BatchedUKF*>(filter.get());
        ukf->setMap(sc.map_xy);
        check(slam_run_sim(ukf->handle(), sc.cmds.data(), T));
        filter->publishState();
        err = ukf->errorStats();
        std::printf("driver ok: run ukf batch=%d L=%d steps=%d M0=%d timestep=%d P_len=%zu X_len=%zu ", B, L, T, ukf->last_state.M,
                    ukf->last_state.timestep, ukf->last_state.P.size(), ukf->last_state.X.size());
    }
    double mean = 0;
    for (double e : err) mean += e;
    std::printf("mean_avg_err=%.9f first_cmd=%.9g,%.9g map0=%.17g,%.17g\n", mean / B, sc.cmds[0], sc.cmds[1], sc.map_xy[0], sc.map_xy[1]);
    return 0;
}

// the global batch over several GPUs of the node from this one process (SURVEY.md section 8(e) "Host")
static int run_multi(const std::string& kind, int64_t B, int L, int T, int gpus, uint64_t seed, int gather_mode) {
    const Scenario sc = make_scenario(seed, L, T);
    slam_config cfg;
    check(slam_config_default(&cfg));
    std::vector<int> devices(gpus);
    for (int i = 0; i < gpus; ++i) devices[i] = i;
    slam_multi* m = nullptr;
    check(slam_multi_create(&cfg, kind == "ekf" ? SLAM_EKF_SLAM : SLAM_UKF_SLAM, B, L, SLAM_F64, devices.data(), gpus, &m));
    check(slam_multi_set_map(m, sc.map_xy.data(), L));
    check(slam_multi_init(m, 0.f, 0.f, 0.f));
    check(slam_multi_run_sim(m, sc.cmds.data(), T));          // every device: get_cmd + Filter::update for its shard, T ticks
    check(slam_multi_sync(m));
    std::vector<double> err((size_t)B);
    check(slam_multi_error_stats(m, err.data(), gather_mode));   // the one gather of the run
    std::vector<int32_t> fl((size_t)B);
    check(slam_multi_status(m, fl.data()));
    double mean = 0;
    int flagged = 0;
    for (int64_t i = 0; i < B; ++i) { mean += err[(size_t)i]; flagged += fl[(size_t)i] != 0; }
    int64_t f0 = 0, c0 = 0;
    check(slam_multi_shard(m, gpus - 1, &f0, &c0));
    std::printf("driver ok: run_multi %s global_batch=%lld gpus=%d gather=%s last_shard=[%lld,+%lld) flagged=%d mean_avg_err=%.9f\n", kind.c_str(),
                (long long)B, gpus, gather_mode ? "rccl" : "host", (long long)f0, (long long)c0, flagged, mean / (double)B);
    slam_multi_destroy(m);
    return 0;
}

// `filter: "pose_graph"` (params.yaml:11): the node runs a secondary filter first and hands its estimate to the pose
// graph every tick (localization_node.cpp:124-131); the pose graph solves when timestep+1 >= num_iterations.
static int run_pose_graph(int B, int L, int T) {
    LocalizationNode node;
    node.filter = std::make_unique<BatchedPoseGraph>(B, /*num_iterations=*/T, L);       // localization_node.cpp:45-46
    node.filter_secondary = std::make_unique<NaiveFilter>();                              // :68-69
    auto* pg = static_cast<BatchedPoseGraph*>(node.filter.get());
    slam_config cfg;
    check(slam_config_default(&cfg));
    node.filter->readParams(cfg); node.filter_secondary->readParams(cfg);
    node.initCallback(0.f, 0.f, 0.f);                                                    // :100-105
    const Scenario sc = make_scenario(7, L, T);
    for (int t = 0; t < T; ++t) {
        auto cmd = std::make_shared<Command>(); cmd->fwd = sc.cmds[2 * t]; cmd->ang = sc.cmds[2 * t + 1];
        auto meas = std::make_shared<Float32MultiArray>();
        if (t % 3 == 0) meas->data = {(float)(t / 30), 2.0f + 0.01f * (float)(t % 7), 0.3f - 0.01f * (float)(t % 5)};   // a landmark seen now and then
        node.cmdCallback(cmd); node.lmMeasCallback(meas);
        node.iterate();
    }
    std::printf("driver ok: pose_graph batch=%d poses=%d solved=%d M0=%d result_topic=%d x_len=%zu conns=%zu\n", B, pg->timestep + 1,
                (int)pg->solved_pose_graph, pg->last_state.M, (int)pg->last_state.is_result, pg->last_state.x_v.size(),
                pg->last_state.meas_connections.size() / 2);
    return 0;
}

int main(int argc, char** argv) {
    try {
        const std::string mode = argc > 1 ? argv[1] : "";
        if (mode == "run_multi" && argc >= 7)
            return run_multi(argv[2], atoll(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), argc > 7 ? strtoull(argv[7], nullptr, 10) : 1234ull,
                             argc > 8 ? atoi(argv[8]) : 0);
        if (mode == "run_multi" && argc >= 7)
            return run_multi(argv[2], atoll(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), argc > 7 ? strtoull(argv[7], nullptr, 10) : 1234ull,
                             argc > 8 ? atoi(argv[8]) : 0);
        if (mode == "stream" && argc >= 7) return run_stream(argv[2], atoi(argv[3]), atoi(argv[4]), argv[5], argv[6]);
        if (mode == "run" && argc >= 6) return run_scenario(argv[2], atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), argc > 6 ? strtoull(argv[6], nullptr, 10) : 1234ull);
        if (mode == "pose_graph" && argc >= 5) return run_pose_graph(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]));
        std::fprintf(stderr, "usage: filter_driver stream <ekf|ukf|ukf_loc> <batch> <L_max> <stream.txt> <dump.bin>\n"
                             "       filter_driver run <ekf|ukf> <batch> <L> <steps> [seed]\n"
                             "       filter_driver run_multi <ekf|ukf> <global batch> <L> <steps> <gpus> [seed] [gather 0|1]\n"
                             "       filter_driver pose_graph <batch> <L> <steps>\n");
        return 2;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "driver failed: %s\n", e.what());
        return 1;
    }
}
