// Compile-time defaults of the reference application (Include/Constants.hpp:4-22).
// In this build they are only defaults: every value is a runtime parameter of the C-ABI.
#pragma once

constexpr auto WIDTH = 1280u; // starting resolution, may change during execution
constexpr auto HEIGHT = 720u;
constexpr auto WIDTHF = static_cast<float>(WIDTH);
constexpr auto HEIGHTF = static_cast<float>(HEIGHT);
constexpr auto FOV = 60.f; // unused by the reference too (Camera.cpp:15 hard-codes 60 * 3.14 / 180)

constexpr auto PATHCOUNT = 1 << 21; // 2M pooled paths
constexpr auto MAX_LIGHTS = 128;
constexpr auto NUM_THREADS = 256;
// The reference sizes its grid for a 34-SM NVIDIA part (NUM_SM * 8 groups) and loops ITERATIONS times per thread;
// integer division leaves 8192 of the 2^21 slots unprocessed.  REFERENCE_LIVE_PATHS reproduces that count.
constexpr auto REFERENCE_GRID_THREADS = 34 * 8 * NUM_THREADS;
constexpr auto REFERENCE_LIVE_PATHS = (PATHCOUNT / REFERENCE_GRID_THREADS) * REFERENCE_GRID_THREADS;

constexpr auto CAPTURE_DIR_NAME = "Captures";
constexpr auto CAPTURE_NAME = "potato";
constexpr auto DEFAULT_SCENE = "bunny_glass/scene.gltf";
