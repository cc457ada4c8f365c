// scenario_capi.cpp — C ABI of the host-side scenario generators (include/slam_scenario.hpp; reference sim_node.py:63-206).
#include "../../include/slam_batch.h"
#include "../../include/slam_scenario.hpp"

#include <string.h>

#include "capi_internal.h"

extern "C" int slam_scenario_make(const char* map_type, const char* fixed_maps_json, uint64_t seed, int num_landmarks,
                                  int num_iterations, double* map_xy, int map_capacity, int32_t* num_landmarks_out, float* cmds) {
    if (!map_type || num_iterations < 0 || !num_landmarks_out) return slam_internal_fail(SLAM_ERR_ARG, "bad argument");
    try {
        const slam_amd::Scenario sc = slam_amd::make_scenario(seed, num_landmarks, num_iterations, map_type, slam_amd::ScenarioOptions(),
                                                              fixed_maps_json ? fixed_maps_json : "");
        const int L = (int)(sc.map_xy.size() / 2);
        *num_landmarks_out = L;
        if (map_xy) {
            if (L > map_capacity) return slam_internal_fail(SLAM_ERR_ARG, "map needs room for %d landmarks, got %d", L, map_capacity);
            memcpy(map_xy, sc.map_xy.data(), sizeof(double) * sc.map_xy.size());
        }
        if (cmds) memcpy(cmds, sc.cmds.data(), sizeof(float) * sc.cmds.size());
    } catch (const std::exception& e) {
        return slam_internal_fail(SLAM_ERR_IO, "%s", e.what());
    }
    return SLAM_OK;
}
