/* stub: shadows the MSVC-only src/utils/cl_exception.hpp (unused by scene.cpp) */
#pragma once
