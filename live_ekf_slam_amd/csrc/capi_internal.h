// capi_internal.h — shared by the translation units that implement the C ABI (not installed, not part of the ABI).
#pragma once
// records the message for slam_last_error() and returns `code`
extern "C" int slam_internal_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// slam_error_stats into a DEVICE buffer of `pad` doubles on the handle's device (entries past the batch are zero), complete when the
// call returns (the handle's stream is synchronised): the send buffer of slam_multi_error_stats' RCCL gather
struct slam_handle;
extern "C" int slam_internal_error_stats_dev(slam_handle* h, double* d_out, long long pad);
