// asan_main.cpp — CPU sanitizer run (AddressSanitizer + UndefinedBehaviorSanitizer) over the oracle and over the product's
// HOST-side code that takes untrusted text or sizes: the params.yaml reader behind slam_config_load
// (live_ekf_slam_amd/csrc/host/config_parse.h), the message-stream reader of filter_driver (host/stream_parse.h) and the
// scenario generators (include/slam_scenario.hpp).  TEST INFRASTRUCTURE, NOT PRODUCT: built by `make -C oracle asan`
// into oracle/_asan/oracle_asan and run by tests/test_sanitizers_cpu.py (-m "not gpu").  GPU AddressSanitizer is not
// available on this pool; the reference's only safety net on this path is eigen_assert -> exception (filter.h:5).
// Exit code 0 = every check ran and no sanitizer report (the binary is built with -fno-sanitize-recover=all, so a report
// aborts).  Usage: oracle_asan <scratch dir> <path to live_ekf_slam_amd/data/fixed_maps.json>
#include <stdio.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include "../include/slam_batch.h"
#include "../include/slam_scenario.hpp"
#include "../live_ekf_slam_amd/csrc/host/config_parse.h"
#include "../live_ekf_slam_amd/csrc/host/stream_parse.h"

extern "C" {
double orc_run_ekf_batch(const slam_config* cfg, int L_max, int math, int mode, const double* map_xy, int L, const float* cmds, int T,
                         uint64_t seed, int64_t inst0, int B, int nthreads, double* x_out, double* P_out, int* M_out, int* ids_out,
                         double* avg_err, int* flags, double* truth_out, int64_t* k_total, const double* vision);
double orc_run_ukf_batch(const slam_config* cfg, int L_max, int math, const double* map_xy, int L, const float* cmds, int T, uint64_t seed,
                         int64_t inst0, int B, int nthreads, double* x_out, double* P_out, int* M_out, int* ids_out, double* avg_err,
                         int* flags, double* truth_out, const double* vision);
double orc_run_pgs_batch(const slam_config* cfg, int L_max, int KP, int math, int lin_mode, const double* map_xy, int L, const float* cmds,
                         int T, uint64_t seed, int64_t inst0, int B, int nthreads, double* pose_init, double* pose_res, double* lm_res,
                         int* M_out, int* ids_out, int* istats, double* dstats, double* avg_err, double* truth_xy, float* meas_out,
                         int* cnt_out);
}

static int g_checks = 0, g_failed = 0;
#define CHECK(cond, what)                                                   \
    do {                                                                    \
        g_checks += 1;                                                      \
        if (!(cond)) { g_failed += 1; fprintf(stderr, "FAILED: %s\n", what); } \
    } while (0)

static void default_cfg(slam_config* c) {   // params.yaml:25-52 (same values as slam_config_default, which lives in the HIP library)
    memset(c, 0, sizeof(*c));
    c->V_00 = 0.01; c->V_11 = 0.001; c->W_00 = 0.01; c->W_11 = 0.01;
    c->landmark_id_is_known = 1; c->min_landmark_separation = 0.1f;
    c->d_max = 0.1; c->th_max = 0.0546; c->range_max = 3.0; c->fov_min = -1.57; c->fov_max = 1.57;
    c->replicate_vw_quirk = 1; c->ukf_float_trig = 1;
}

static std::string write_file(const std::string& dir, const char* name, const std::string& text) {
    const std::string p = dir + "/" + name;
    FILE* f = fopen(p.c_str(), "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", p.c_str()); exit(2); }
    fwrite(text.data(), 1, text.size(), f);
    fclose(f);
    return p;
}

static void config_checks(const std::string& dir) {
    slam_config c;
    std::string err;
    default_cfg(&c);
    const std::string good =
        "init_pose:\n  x: 1.5\n  y: -2.0\n  yaw: 0.25\nconstraints:\n  commands:\n    d_max: 0.2\n    th_max: 0.1\n"
        "  vision:\n    range_max: 4.0\n    fov_min: -1.0\n    fov_max: 1.0\n  measurements:\n    landmark_id_is_known: false\n"
        "    min_landmark_separation: 0.3\nmap:\n  min_landmark_separation: 0.05\nprocess_noise:\n  mean:\n    v_d: 0.01\n    v_th: 0.0\n"
        "  cov:\n    V_00: 0.02\n    V_11: 0.002\nsensing_noise:\n  mean:\n    w_r: 0.0\n    w_b: 0.0\n  cov:\n    W_00: 0.03\n    W_11: 0.04\n";
    CHECK(slam_host::config_parse_file(&c, write_file(dir, "good.yaml", good).c_str(), &err) == 0, "good yaml parses");
    CHECK(c.init_x == 1.5 && c.init_y == -2.0 && c.init_yaw == 0.25 && c.d_max == 0.2 && c.range_max == 4.0, "good yaml: poses / constraints");
    CHECK(c.landmark_id_is_known == 0 && c.min_landmark_separation == 0.3f && c.V_00 == 0.02 && c.W_11 == 0.04 && c.v_d == 0.01f,
          "good yaml: noise / measurement keys (map.min_landmark_separation ignored)");
    CHECK(slam_host::config_parse_file(&c, (dir + "/does_not_exist.yaml").c_str(), &err) == 1, "missing file reported");
    CHECK(slam_host::config_parse_file(&c, dir.c_str(), &err) != 2, "a directory instead of a file does not crash");
    // malformed inputs: each must return without a sanitizer report; the config keeps finite values
    std::string longline(5000, 'x');
    longline += ": 1\nV_00: 0.5\n";
    std::string longkey = std::string(300, 'k') + ": 3\nV_11: 0.25\n";
    std::string binary;
    for (int i = 0; i < 4096; ++i) binary.push_back((char)((i * 73 + 11) & 0xff));
    std::string nul = "V_00: 0.125\n"; nul.push_back('\0'); nul += "W_00: 9\n";
    const char* names[] = {"empty.yaml", "long.yaml", "longkey.yaml", "binary.yaml", "nocolon.yaml", "nonum.yaml", "nul.yaml", "nonl.yaml"};
    const std::string texts[] = {"", longline, longkey, binary, "V_00 0.7\n  v_d\n:::\n: 5\n", "V_00: abc\nv_d:\nW_00: -\n", nul, "V_11: 0.75"};
    for (int i = 0; i < 8; ++i) {
        default_cfg(&c);
        const int rc = slam_host::config_parse_file(&c, write_file(dir, names[i], texts[i]).c_str(), &err);
        CHECK(rc == 0, names[i]);
        CHECK(isfinite(c.V_00) && isfinite(c.V_11) && isfinite(c.W_00), "values stay finite");
        if (i == 1) CHECK(c.V_00 == 0.5, "the line after an over-long line is still read");
        if (i == 7) CHECK(c.V_11 == 0.75, "last line without a newline is read");
    }
    // out-of-range values are rejected, not converted (float / int casts of out-of-range doubles are UB)
    const char* bad[] = {"v_d: 1e300\n", "landmark_id_is_known: 1e30\n", "V_00: inf\n", "w_r: nan\n", "min_landmark_separation: 5\nconstraints:\n  min_landmark_separation: -1e99\n"};
    for (int i = 0; i < 5; ++i) {
        default_cfg(&c);
        CHECK(slam_host::config_parse_file(&c, write_file(dir, "bad.yaml", bad[i]).c_str(), &err) == 2, bad[i]);
    }
}

static void stream_checks() {
    slam_host::StreamLine sl;
    std::string err;
    CHECK(slam_host::parse_stream_line("0.1 -0.05 2 3 1.5 0.25 7 2.0 -0.5", &sl, &err) && !sl.is_map && sl.data.size() == 6 && sl.fwd == 0.1f && sl.data[3] == 7.f, "tick line");
    CHECK(slam_host::parse_stream_line("0.1 0.0 0", &sl, &err) && sl.data.empty(), "tick without detections");
    CHECK(slam_host::parse_stream_line("map 2 0 1.0 2.0 1 -3.0 4.0", &sl, &err) && sl.is_map && sl.data.size() == 6, "map line");
    const char* bad[] = {"", "   ", "0.1", "0.1 0.2", "0.1 0.2 -1", "0.1 0.2 3 1 2 3", "0.1 0.2 1 1 2 3 4", "0.1 0.2 99999999999 1 2 3", "0.1 0.2 1 1 nan 3",
                         "abc 0.2 0", "0.1x 0.2 0", "map", "map -5", "map 5000", "map 1 0 1", "0.1 0.2 2147483648", "1e99 0 0", "0.1 0.2 1.5 1 2 3"};
    for (const char* b : bad) CHECK(!slam_host::parse_stream_line(b, &sl, &err), b);
}

static void scenario_checks(const std::string& fixed_maps) {
    using namespace slam_amd;
    for (const char* mt : {"random", "grid", "demo", "igvc1"}) {
        Scenario sc = make_scenario(5, 20, 120, mt, ScenarioOptions(), fixed_maps);
        CHECK(sc.map_xy.size() >= 2 && sc.cmds.size() == 240, mt);
    }
    bool threw = false;
    try { make_scenario(5, 20, 10, "nope"); } catch (const std::exception&) { threw = true; }
    CHECK(threw, "unknown map type throws");
    threw = false;
    try { make_scenario(5, 20, 10, "demo", ScenarioOptions(), "/nonexistent.json"); } catch (const std::exception&) { threw = true; }
    CHECK(threw, "missing fixed-map file throws");
    threw = false;
    try { make_scenario(5, 0, 10, "random"); } catch (const std::exception&) { threw = true; }
    CHECK(threw, "a scenario without landmarks throws (the planner has nowhere to go)");
    Scenario one = make_scenario(7, 1, 30, "random");
    CHECK(one.map_xy.size() == 2 && one.cmds.size() == 60, "one landmark");
    Scenario none = make_scenario(7, 3, 0, "random");
    CHECK(none.cmds.empty(), "zero iterations");
}

static void oracle_checks() {
    slam_config c;
    default_cfg(&c);
    const int L = 12, T = 90, B = 3;
    slam_amd::Scenario sc = slam_amd::make_scenario(1234, L, T);
    std::vector<double> vis((size_t)3 * T);
    for (int t = 0; t < T; ++t) { vis[3 * t] = t == 0 ? 1e9 : (t > 40 && t < 50 ? 1e-6 : 3.0); vis[3 * t + 1] = t == 0 ? -4.0 : -1.57; vis[3 * t + 2] = t == 0 ? 4.0 : 1.57; }
    for (int mode = 0; mode < 4; ++mode)   // MODE_FAST / MODE_DENSE, with and without fp32 storage (bit 2)
        for (int math = 0; math < 2; ++math) {
            const int nm = 3 + 2 * L;
            std::vector<double> x((size_t)B * nm), P((size_t)B * nm * nm), err(B), truth(3 * B);
            std::vector<int> M(B), ids((size_t)B * L), fl(B);
            int64_t kt = 0;
            orc_run_ekf_batch(&c, L, math, mode, sc.map_xy.data(), L, sc.cmds.data(), T, 11, 5, B, 2, x.data(), P.data(), M.data(), ids.data(),
                              err.data(), fl.data(), truth.data(), &kt, vis.data());
            CHECK(M[0] == L && fl[0] == 0 && isfinite(x[0]) && isfinite(err[B - 1]) && kt > 0, "EKF oracle run");
        }
    {   // capacity smaller than the map, unknown ids, quirk off: the edge branches of EKF::update
        slam_config c2 = c; c2.landmark_id_is_known = 0; c2.replicate_vw_quirk = 0;
        const int Ls = 5, nm = 3 + 2 * Ls;
        std::vector<double> x((size_t)B * nm), P((size_t)B * nm * nm);
        std::vector<int> M(B), fl(B);
        orc_run_ekf_batch(&c2, Ls, 1, 0, sc.map_xy.data(), L, sc.cmds.data(), T, 11, 5, B, 1, x.data(), P.data(), M.data(), nullptr, nullptr,
                          fl.data(), nullptr, nullptr, vis.data());
        CHECK(M[0] <= Ls, "EKF oracle: capacity respected");
    }
    {
        const int nm = 4 + 2 * L;
        std::vector<double> x((size_t)B * nm), P((size_t)B * nm * nm), err(B);
        std::vector<int> M(B), ids((size_t)B * L), fl(B);
        orc_run_ukf_batch(&c, L, 1, sc.map_xy.data(), L, sc.cmds.data(), 50, 11, 5, B, 2, x.data(), P.data(), M.data(), ids.data(), err.data(),
                          fl.data(), nullptr, vis.data());
        CHECK(M[0] == L && isfinite(x[0]) && isfinite(P[0]), "UKF oracle run");
    }
    {
        const int Tp = 60, KP = 8, Bp = 2;
        std::vector<double> pi((size_t)Bp * (Tp + 1) * 3), pr((size_t)Bp * (Tp + 1) * 3), lmr((size_t)Bp * L * 2), ds((size_t)Bp * 3), ae((size_t)Bp * 2);
        std::vector<int> M(Bp), ids((size_t)Bp * L), is((size_t)Bp * 3);
        orc_run_pgs_batch(&c, L, KP, 1, 0, sc.map_xy.data(), L, sc.cmds.data(), Tp, 11, 5, Bp, 1, pi.data(), pr.data(), lmr.data(), M.data(),
                          ids.data(), is.data(), ds.data(), ae.data(), nullptr, nullptr, nullptr);
        CHECK(is[0] > 0 && isfinite(pr[3 * Tp]) && ds[1] <= ds[0], "pose-graph oracle solve");
    }
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: oracle_asan <scratch dir> <fixed_maps.json>\n"); return 2; }
    config_checks(argv[1]);
    stream_checks();
    scenario_checks(argv[2]);
    oracle_checks();
    printf("sanitizer run: %d checks, %d failed\n", g_checks, g_failed);
    return g_failed ? 1 : 0;
}
