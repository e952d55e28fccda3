// vdk_host.h — host-side plumbing shared by the C-ABI entry points: error codes, thread-local
// last-error string, launch checking.  Every entry point returns 0 or a negative code; it never
// allocates, never synchronises and borrows all pointers (SURVEY.md §8(b) contract).
#pragma once
#include <stdint.h>
#include <stddef.h>

#include "visiondk.h"

int vdk_fail(int code, const char* msg);
int vdk_check_launch(const char* what);
