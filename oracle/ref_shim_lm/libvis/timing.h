#pragma once
#include <chrono>
namespace vis { class Timer { public: explicit Timer(const char*) : t0(std::chrono::steady_clock::now()) {} double Stop(bool = true) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } private: std::chrono::steady_clock::time_point t0; }; }
