/* stub: the reference's scene.cpp includes "render.hpp" without using it (scene.cpp:32) */
#pragma once
