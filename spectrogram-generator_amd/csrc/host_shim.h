// Declarations shared by host_shim.cpp (no HIP) and the .hip translation units of libspectro.so.
#pragma once
#include "spectro.h"

namespace sg {

// thread-local error plumbing -------------------------------------------------
void set_error(const char* fmt, ...);
const char* last_error_cstr();
// argument triage of sg_plan_create (everything that needs no device): SG_OK or SG_ERR_ARG with the message set
int check_plan_args(int nperseg, int nfft, int hop, int detrend, double fs, int scaling, int mode, int dtype);

}  // namespace sg
