// capi_internal.h — shared by the translation units that implement the C ABI (not installed, not part of the ABI).
#pragma once
// records the message for slam_last_error() and returns `code`
extern "C" int slam_internal_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
