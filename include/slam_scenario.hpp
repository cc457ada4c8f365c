// slam_scenario.hpp — host-side C++ scenario generators of the reference simulator, so that a C++ host can run a BASELINE
// configuration without Python:
//   generate_landmarks        ekf_ws/src/base_pkg/src/sim_node.py:155-206  (map_type random | grid | demo | igvc1)
//   generate_full_trajectory  ekf_ws/src/base_pkg/src/sim_node.py:63-138   (nearest-neighbour tour + command clamps)
// The reference draws from CPython's global Mersenne Twister (`from random import random`, sim_node.py:16); PyRandom below is
// that generator (MT19937, CPython's init_by_array seeding of an int, 53-bit random()), so a seed reproduces the reference's
// scenario bit for bit: tests/test_scenario.py compares map and commands with fixtures captured from the imported reference
// for all four map types.  Arithmetic mirrors the Python expressions operation by operation, including `x ** 2` and
// `s ** (1/2)` as libm pow (what CPython's float_pow calls), math.remainder and np.sign.
// Only the blank occupancy map is supported (every cell free), i.e. no rejection of landmarks by obstacles.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

namespace slam_amd {

// CPython `random.Random(seed)` for a non-negative int seed: _randommodule.c random_seed -> init_by_array(32-bit words of the
// seed, least significant first); random() = (genrand_uint32 >> 5, >> 6) -> (a * 2^26 + b) / 2^53.
class PyRandom {
public:
    explicit PyRandom(uint64_t seed) {
        uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
        init_by_array(key, key[1] ? 2 : 1);
    }
    double random() {
        const uint32_t a = next() >> 5, b = next() >> 6;
        return ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0);
    }

private:
    static constexpr int N = 624, Mm = 397;
    uint32_t mt[N];
    int idx = N + 1;
    void init_genrand(uint32_t s) {
        mt[0] = s;
        for (int i = 1; i < N; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = N;
    }
    void init_by_array(const uint32_t* key, int len) {
        init_genrand(19650218u);
        int i = 1, j = 0;
        for (int k = (N > len ? N : len); k; --k) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
            if (++i >= N) { mt[0] = mt[N - 1]; i = 1; }
            if (++j >= len) j = 0;
        }
        for (int k = N - 1; k; --k) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
            if (++i >= N) { mt[0] = mt[N - 1]; i = 1; }
        }
        mt[0] = 0x80000000u;
    }
    uint32_t next() {
        if (idx >= N) {
            for (int k = 0; k < N; ++k) {
                const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % N] & 0x7fffffffu);
                mt[k] = mt[(k + Mm) % N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return y;
    }
};

struct ScenarioOptions {   // params.yaml:68-72, 88-91, 25-28, 15 (reference defaults)
    double bound = 10.0, min_landmark_separation = 0.05, grid_step = 4.0, landmark_noise = 0.2, visitation_threshold = 3.0,
           d_max = 0.1, th_max = 0.0546, display_region_mult = 1.0;
};

inline double py_norm(double ax, double ay, double bx, double by) {   // sim_node.py:35-38
    return std::pow(std::pow(ax - bx, 2.0) + std::pow(ay - by, 2.0), 0.5);
}

// the fixed maps (demo_map sim_node.py:26-30, igvc1 barrels :190) are data: live_ekf_slam_amd/data/fixed_maps.json
inline std::vector<double> load_fixed_map(const std::string& json_path, const std::string& name) {
    FILE* f = std::fopen(json_path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + json_path);
    std::string txt;
    char buf[4096];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0) txt.append(buf, got);
    std::fclose(f);
    size_t p = txt.find("\"" + name + "\"");
    if (p == std::string::npos) throw std::runtime_error("no map '" + name + "' in " + json_path);
    p = txt.find('[', p);
    if (p == std::string::npos) throw std::runtime_error("malformed map '" + name + "' in " + json_path);
    std::vector<double> out;
    int depth = 0;
    for (size_t i = p; i < txt.size(); ++i) {
        const char c = txt[i];
        if (c == '[') ++depth;
        else if (c == ']') { if (--depth == 0) break; }
        else if (c == '-' || (c >= '0' && c <= '9')) {
            char* end = nullptr;
            out.push_back(std::strtod(txt.c_str() + i, &end));
            i = (size_t)(end - txt.c_str()) - 1;
        }
    }
    if (out.size() % 2) throw std::runtime_error("malformed map '" + name + "'");
    return out;
}

// Landmark map [L][2] (x, y), id = row (sim_node.py:155-206).  `num_landmarks` is ignored for grid and the fixed maps.
inline std::vector<double> generate_landmarks(const std::string& map_type, int num_landmarks, PyRandom& rng,
                                              const ScenarioOptions& o = ScenarioOptions(), const std::string& fixed_maps_json = "") {
    std::vector<double> lm;
    if (map_type == "demo" || map_type == "igvc1") return load_fixed_map(fixed_maps_json, map_type);
    if (map_type == "random" || map_type == "rand") {
        // rejection sampling cannot place more landmarks than the region holds at the minimum separation: bound the attempts
        long attempts = 0;
        while ((int)(lm.size() / 2) < num_landmarks) {
            if (++attempts > 1000L * (num_landmarks > 0 ? num_landmarks : 1) + 100000L)
                throw std::runtime_error("generate_landmarks: cannot place that many landmarks at this separation");
            const double x = 2 * o.bound * rng.random() - o.bound;     // sim_node.py:179 (x drawn first)
            const double y = 2 * o.bound * rng.random() - o.bound;
            bool close = false;
            for (size_t i = 0; i < lm.size() && !close; i += 2) close = py_norm(lm[i], lm[i + 1], x, y) < o.min_landmark_separation;
            if (close) continue;                                       // :183
            lm.push_back(x); lm.push_back(y);
        }
        return lm;
    }
    if (map_type == "grid") {                                          // :167-174, np.arange(-bound + shift, bound, step)
        const double start = -o.bound + o.grid_step / 2;
        const int cnt = (int)std::ceil((o.bound - start) / o.grid_step);
        for (int r = 0; r < cnt; ++r)
            for (int c = 0; c < cnt; ++c) { lm.push_back(start + r * o.grid_step); lm.push_back(start + c * o.grid_step); }
        return lm;
    }
    throw std::runtime_error("unsupported map_type " + map_type + " (random | grid | demo | igvc1)");
}

// Commands [T][2] (fwd, ang) as doubles (sim_node.py:63-138); the caller rounds to float32 for the Command wire format.
inline std::vector<double> generate_full_trajectory(const std::vector<double>& landmarks, int num_iterations, PyRandom& rng,
                                                    const ScenarioOptions& o = ScenarioOptions(), double x0 = 0.0, double y0 = 0.0,
                                                    double yaw0 = 0.0) {
    const int L = (int)(landmarks.size() / 2);
    // the reference's planner indexes its first landmark unconditionally (sim_node.py:90-94: IndexError on an empty map)
    if (L <= 0) throw std::runtime_error("generate_full_trajectory needs at least one landmark");
    if (num_iterations < 0) throw std::runtime_error("num_iterations must not be negative");
    const double lo = -o.bound * o.display_region_mult + 1, hi = o.bound * o.display_region_mult - 1;
    std::vector<double> rough(2 * L);
    for (int i = 0; i < L; ++i) {   // planner's noisy copy of the map: 2 draws per landmark, x then y (:82-87)
        const double nx = landmarks[2 * i] + 2 * o.landmark_noise * rng.random() - o.landmark_noise;
        const double ny = landmarks[2 * i + 1] + 2 * o.landmark_noise * rng.random() - o.landmark_noise;
        rough[2 * i] = std::fmax(lo, std::fmin(nx, hi));
        rough[2 * i + 1] = std::fmax(lo, std::fmin(ny, hi));
    }
    double pose[3] = {x0, y0, yaw0};
    int start = 0;
    double best = py_norm(rough[0], rough[1], pose[0], pose[1]);
    for (int i = 0; i < L; ++i) {
        const double d = py_norm(rough[2 * i], rough[2 * i + 1], pose[0], pose[1]);
        if (d < best) { start = i; best = d; }
    }
    std::vector<int> tour{start}, todo;
    for (int i = 0; i < L; ++i) if (i != start) todo.push_back(i);
    int cur = start;
    while (!todo.empty()) {          // nearest-neighbour tour (:96-110)
        int nxt = -1; double nd = -1.0; size_t at = 0;
        for (size_t q = 0; q < todo.size(); ++q) {
            const int i = todo[q];
            const double dd = py_norm(rough[2 * i], rough[2 * i + 1], rough[2 * cur], rough[2 * cur + 1]);
            if (nd < 0 || dd < nd) { nxt = i; nd = dd; at = q; }
        }
        tour.push_back(nxt); todo.erase(todo.begin() + at); cur = nxt;
    }
    const double tau = 2 * 3.14159265358979323846;   // math.tau
    std::vector<double> cmds(2 * (size_t)num_iterations);
    size_t head = 0;                                  // tour[head] is the current goal; arrived goals rotate to the end
    for (int t = 0; t < num_iterations; ++t) {
        if (py_norm(pose[0], pose[1], rough[2 * tour[head]], rough[2 * tour[head] + 1]) < o.visitation_threshold) head = (head + 1) % tour.size();
        const double gx = rough[2 * tour[head]], gy = rough[2 * tour[head] + 1];
        double d = std::fmin(py_norm(gx, gy, pose[0], pose[1]), o.d_max);
        double hdg = std::remainder(std::atan2(gy - pose[1], gx - pose[0]) - pose[2], tau);
        if (std::fabs(hdg) > o.th_max) hdg = o.th_max * (hdg > 0 ? 1.0 : (hdg < 0 ? -1.0 : 0.0));   // th_max * np.sign(hdg)
        const double c = std::cos(pose[2]), s = std::sin(pose[2]);
        pose[0] = pose[0] + d * c; pose[1] = pose[1] + d * s; pose[2] = pose[2] + hdg;
        cmds[2 * (size_t)t] = d; cmds[2 * (size_t)t + 1] = hdg;
    }
    return cmds;
}

struct Scenario {
    std::vector<double> map_xy;   // [L][2]
    std::vector<float> cmds;      // [T][2] float32 (Command.msg)
};

// Map + command sequence for one scenario seed, the reference's draw order: map first, then the planner.
inline Scenario make_scenario(uint64_t seed, int num_landmarks, int num_iterations, const std::string& map_type = "random",
                              const ScenarioOptions& o = ScenarioOptions(), const std::string& fixed_maps_json = "") {
    PyRandom rng(seed);
    Scenario sc;
    sc.map_xy = generate_landmarks(map_type, num_landmarks, rng, o, fixed_maps_json);
    const std::vector<double> c = generate_full_trajectory(sc.map_xy, num_iterations, rng, o);
    sc.cmds.resize(c.size());
    for (size_t i = 0; i < c.size(); ++i) sc.cmds[i] = (float)c[i];
    return sc;
}

}  // namespace slam_amd
