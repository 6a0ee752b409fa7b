/* stub: shared_structures.h:30 includes <CL/cl.h>; only CL_FLT_MAX is used (mathlib.cpp:89-90) */
#pragma once
#define CL_FLT_MAX 340282346638528859811704183484516925440.0f
