// Per-thread last-error text behind pt_last_error().
#include <string>

#include "pt_host.h"

namespace {
thread_local std::string g_last_error;
}

namespace pt {
void set_error(const char* msg) { g_last_error = msg ? msg : ""; }
}  // namespace pt

extern "C" const char* pt_last_error(void) { return g_last_error.c_str(); }
