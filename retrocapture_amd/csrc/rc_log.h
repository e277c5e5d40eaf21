// Minimal logger with the reference's convention: functions return bool / handles and
// describe failures through the log (reference src/utils/Logger.h:18-21; level from the
// RETROCAPTURE_LOG_LEVEL environment variable, Logger.cpp:41-44).  The last error text is
// also kept per thread so the C ABI can hand it to a caller (rc_last_error()).
#pragma once
#include <string>

namespace rc {
enum class LogLevel { Debug = 0, Info = 1, Warn = 2, Error = 3 };
void log(LogLevel level, const std::string& msg);
bool log_enabled(LogLevel level);   // whether a message of this level would be printed (to skip building it)
const std::string& last_error();
void clear_last_error();
// While an engine that was told shader sources may be absent (setAllowMissingSources) loads a preset, the
// per-file "not found" warnings are counted instead of printed; the engine prints one summary line.
struct MissingSourceScope {
  explicit MissingSourceScope(bool quiet);
  ~MissingSourceScope();
  int count() const;
};
// returns true if the warning was absorbed by an active scope
bool note_missing_source();  // the C ABI clears it on entry of calls whose status depends on it
}  // namespace rc

#define RC_LOG_DEBUG(m) ::rc::log(::rc::LogLevel::Debug, (m))
#define RC_LOG_INFO(m) ::rc::log(::rc::LogLevel::Info, (m))
#define RC_LOG_WARN(m) ::rc::log(::rc::LogLevel::Warn, (m))
#define RC_LOG_ERROR(m) ::rc::log(::rc::LogLevel::Error, (m))
