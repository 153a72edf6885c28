// capi_common.h — what every capi_<kernel>.hip includes: the internal declarations, the host entries' transfer pipeline
// (host_pipeline.h) and the multi-device layer on top of it (host_multi.h).
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "gbx_internal.h"

namespace gbx {
int require_device();      // GBX_OK, or GBX_ERR_NO_DEVICE with the error text set (gbx_core.hip)
}

#include "host_pipeline.h"
#include "host_multi.h"
#include "host_combine.h"
