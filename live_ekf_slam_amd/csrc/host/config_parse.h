// config_parse.h — the `key: value` reader behind slam_config_load (include/slam_batch.h): replaces YAML::LoadFile +
// Filter::readCommonParams (localization_node.cpp:29-30, filter.h:105-121) for a params.yaml-shaped file.  Host-only, no HIP:
// slam_capi.cpp includes it for the product, oracle/asan_main.cpp compiles it under ASan + UBSan and feeds it malformed files
// (the text is untrusted input).  Unknown keys are ignored, missing keys keep their defaults, over-long lines are skipped
// whole, values outside the target type's range are rejected instead of converted (float/int casts of out-of-range doubles
// are undefined behaviour).
#pragma once
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "../../../include/slam_batch.h"

namespace slam_host {

// matches "<spaces>key: value [# comment]"; false if the line is about another key or holds no number
inline bool parse_scalar(const char* line, const char* key, double* out) {
    const char* p = line;
    while (*p == ' ' || *p == '\t') ++p;
    const size_t kl = strlen(key);
    if (strncmp(p, key, kl) != 0 || p[kl] != ':') return false;
    p += kl + 1;
    while (*p == ' ' || *p == '\t') ++p;
    if (strncmp(p, "true", 4) == 0) { *out = 1.0; return true; }
    if (strncmp(p, "false", 5) == 0) { *out = 0.0; return true; }
    char* end = nullptr;
    const double v = strtod(p, &end);
    if (end == p) return false;
    *out = v;
    return true;
}

inline bool fits_float(double v) { return isfinite(v) && fabs(v) <= 3.0e38; }
inline bool fits_int(double v) { return isfinite(v) && fabs(v) <= 2.0e9; }

// 0 = ok, 1 = cannot open, 2 = a value does not fit its field (err names the key)
inline int config_parse_file(slam_config* c, const char* path, std::string* err) {
    FILE* f = fopen(path, "r");
    if (!f) { if (err) *err = std::string("cannot open ") + path; return 1; }
    char line[1024];
    std::string section;
    double v;
    int rc = 0;
    auto as_float = [&](const char* key, float* dst) {
        if (!fits_float(v)) { rc = 2; if (err) *err = std::string("value of ") + key + " does not fit a float"; return; }
        *dst = (float)v;
    };
    auto as_double = [&](const char* key, double* dst) {
        if (!isfinite(v)) { rc = 2; if (err) *err = std::string("value of ") + key + " is not finite"; return; }
        *dst = v;
    };
    while (rc == 0 && fgets(line, sizeof(line), f)) {
        const size_t len = strlen(line);
        if (len == sizeof(line) - 1 && line[len - 1] != '\n') {   // over-long line: drop all of it (its tail must not parse as a new line)
            int ch;
            while ((ch = fgetc(f)) != EOF && ch != '\n') {}
            continue;
        }
        if (line[0] != ' ' && line[0] != '\t' && line[0] != '#' && line[0] != '\n' && line[0] != '\r') {  // top-level key
            char key[128];
            if (sscanf(line, "%127[^:\n]:", key) == 1) section = key;
        }
        if (parse_scalar(line, "v_d", &v)) as_float("v_d", &c->v_d);
        else if (parse_scalar(line, "v_th", &v)) as_float("v_th", &c->v_th);
        else if (parse_scalar(line, "V_00", &v)) as_double("V_00", &c->V_00);
        else if (parse_scalar(line, "V_11", &v)) as_double("V_11", &c->V_11);
        else if (parse_scalar(line, "w_r", &v)) as_float("w_r", &c->w_r);
        else if (parse_scalar(line, "w_b", &v)) as_float("w_b", &c->w_b);
        else if (parse_scalar(line, "W_00", &v)) as_double("W_00", &c->W_00);
        else if (parse_scalar(line, "W_11", &v)) as_double("W_11", &c->W_11);
        else if (parse_scalar(line, "landmark_id_is_known", &v)) {
            if (!fits_int(v)) { rc = 2; if (err) *err = "value of landmark_id_is_known does not fit an int"; }
            else c->landmark_id_is_known = (int)v;
        }
        else if (parse_scalar(line, "min_landmark_separation", &v)) { if (section == "constraints") as_float("min_landmark_separation", &c->min_landmark_separation); }
        else if (parse_scalar(line, "d_max", &v)) as_double("d_max", &c->d_max);
        else if (parse_scalar(line, "th_max", &v)) as_double("th_max", &c->th_max);
        else if (parse_scalar(line, "range_max", &v)) as_double("range_max", &c->range_max);
        else if (parse_scalar(line, "fov_min", &v)) as_double("fov_min", &c->fov_min);
        else if (parse_scalar(line, "fov_max", &v)) as_double("fov_max", &c->fov_max);
        else if (section == "init_pose" && parse_scalar(line, "x", &v)) as_double("init_pose.x", &c->init_x);
        else if (section == "init_pose" && parse_scalar(line, "y", &v)) as_double("init_pose.y", &c->init_y);
        else if (section == "init_pose" && parse_scalar(line, "yaw", &v)) as_double("init_pose.yaw", &c->init_yaw);
    }
    fclose(f);
    return rc;
}

}  // namespace slam_host
