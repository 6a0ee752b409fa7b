// kernels.h -- all device code of the hot path (see kernels_common.h for the data layout).
#pragma once
#include "kernels_common.h"
#include "raygen_kernels.h"
#include "trace_kernels.h"
#include "shade_kernels.h"
#include "frame_kernels.h"
#include "aov_kernels.h"
#include "relayout_kernels.h"
