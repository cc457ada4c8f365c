// lds_attr.h — dynamic LDS beyond the default 64 KiB limit (gfx950: 160 KiB per CU), set correctly for a multi-device, multi-threaded host.
//
// hipFuncAttributeMaxDynamicSharedMemorySize belongs to a (function, device) pair, not to a handle or a launch:
// * once per process (std::call_once) leaves every further device of a single-process multi-GPU host (include/slam_multi.h) at 64 KiB;
// * per launch with the current handle's size lets two handles of different capacity, driven from two host threads, interleave
//   set / launch so that the larger one launches below its request.
// So: once per (function, current device), to the MOST the function can ever ask for ON THAT DEVICE - what the device reports as its
// LDS per workgroup (gfx950: 160 KiB; a 64 KiB device keeps running the launches that fit) minus the function's static LDS,
// as the runtime reports it - and the result is checked.  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <set>
#include <utility>

inline hipError_t slam_allow_full_lds(const void* fn) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({fn, dev})) return hipSuccess;
    hipFuncAttributes a;
    e = hipFuncGetAttributes(&a, fn);
    if (e != hipSuccess) return e;
    int cap = 0;
    if (hipDeviceGetAttribute(&cap, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || cap <= 0) cap = 64 * 1024;
    if (cap > 160 * 1024) cap = 160 * 1024;
    const int most = cap - (int)a.sharedSizeBytes;
    if (most > 64 * 1024 - (int)a.sharedSizeBytes) {   // (at or below the default limit there is nothing to raise)
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, most);
        if (e != hipSuccess) return e;
    }
    done.insert({fn, dev});
    return hipSuccess;
}
