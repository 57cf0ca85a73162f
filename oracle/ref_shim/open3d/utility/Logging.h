// Stand-in for open3d/utility/Logging.h: LogError throws std::runtime_error
// (Logging.h:89-94); the other levels are silent. Format arguments are
// ignored (no fmt in this environment).
#pragma once
#include <stdexcept>
#include <string>
#include "open3d/Macro.h"
namespace open3d {
namespace utility {
template <typename... Args>
[[noreturn]] inline void LogError(const char* format, Args&&...) {
    throw std::runtime_error(std::string("[Open3D Error] ") + format);
}
class Logger {
public:
    template <typename... Args>
    [[noreturn]] static void LogError_(const char*, int, const char*,
                                       const char* format, Args&&...) {
        throw std::runtime_error(std::string("[Open3D Error] ") + format);
    }
};
template <typename... Args> inline void LogWarning(const char*, Args&&...) {}
template <typename... Args> inline void LogInfo(const char*, Args&&...) {}
template <typename... Args> inline void LogDebug(const char*, Args&&...) {}
}  // namespace utility
}  // namespace open3d
