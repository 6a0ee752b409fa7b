/* force-included before the reference's host sources: headers they rely on
 * MSVC to pull in transitively, and std:: spellings libstdc++ does not ship. */
#pragma once
#include <cstdint>
#include <cstring>
#include <cmath>
#include <limits>
#include <cassert>
#include <string>
#include <stdexcept>
namespace std
{
inline float powf(float a, float b) { return ::powf(a, b); }
inline float cosf(float a) { return ::cosf(a); }
inline float sinf(float a) { return ::sinf(a); }
}
