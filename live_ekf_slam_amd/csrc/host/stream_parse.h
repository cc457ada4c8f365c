// stream_parse.h — one line of the recorded message stream filter_driver replays (`filter_driver stream ...`): either
//   map <L> {id x y}*L                      the simulator's true map (trueMapCallback, localization_node.cpp:152-156)
//   <fwd> <ang> <k> {id range bearing}*k    one tick: Command + Float32MultiArray (cmdCallback / lmMeasCallback)
// The file is untrusted text: counts are bounded, every value must parse as a finite float, trailing garbage and short
// lines are errors.  Host-only; compiled under ASan + UBSan by oracle/asan_main.cpp.
#pragma once
#include <math.h>

#include <sstream>
#include <string>
#include <vector>

namespace slam_host {

struct StreamLine {
    bool is_map = false;
    float fwd = 0.f, ang = 0.f;
    std::vector<float> data;   // 3 floats per landmark / detection
};

static constexpr int kStreamMaxItems = 4096;   // landmarks of a map line / detections of a tick

// true = parsed; false = malformed (err says why).  Empty and comment lines are not handled here.
inline bool parse_stream_line(const std::string& line, StreamLine* out, std::string* err) {
    std::istringstream ls(line);
    std::string first;
    if (!(ls >> first)) { if (err) *err = "empty line"; return false; }
    auto read_floats = [&](long n) -> bool {
        out->data.assign((size_t)3 * (size_t)n, 0.f);
        for (auto& v : out->data) {
            if (!(ls >> v) || !isfinite(v)) { if (err) *err = "missing or non-finite value"; return false; }
        }
        std::string rest;
        if (ls >> rest) { if (err) *err = "trailing text: " + rest.substr(0, 32); return false; }
        return true;
    };
    long n = -1;
    if (first == "map") {
        out->is_map = true;
        if (!(ls >> n) || n < 0 || n > kStreamMaxItems) { if (err) *err = "bad landmark count"; return false; }
        return read_floats(n);
    }
    out->is_map = false;
    {
        std::istringstream fs(first);
        if (!(fs >> out->fwd) || !isfinite(out->fwd) || fs.peek() != std::istringstream::traits_type::eof()) { if (err) *err = "bad fwd"; return false; }
    }
    if (!(ls >> out->ang) || !isfinite(out->ang)) { if (err) *err = "bad ang"; return false; }
    if (!(ls >> n) || n < 0 || n > kStreamMaxItems) { if (err) *err = "bad detection count"; return false; }
    return read_floats(n);
}

}  // namespace slam_host
