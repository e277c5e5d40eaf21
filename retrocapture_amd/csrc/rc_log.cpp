#include "rc_log.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>

namespace rc {
namespace {
LogLevel threshold() {
  static LogLevel lvl = [] {
    const char* e = std::getenv("RETROCAPTURE_LOG_LEVEL");
    std::string s = e ? e : "warn";
    std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char)std::tolower(c); });
    if (s == "debug") return LogLevel::Debug;
    if (s == "info") return LogLevel::Info;
    if (s == "error") return LogLevel::Error;
    return LogLevel::Warn;
  }();
  return lvl;
}
thread_local std::string t_last_error;
thread_local int t_missing_depth = 0, t_missing_count = 0;
std::mutex g_io;
}  // namespace

bool log_enabled(LogLevel level) { return level >= threshold(); }

void log(LogLevel level, const std::string& msg) {
  if (level == LogLevel::Error) t_last_error = msg;
  if (level < threshold()) return;
  static const char* names[] = {"DEBUG", "INFO", "WARN", "ERROR"};
  std::lock_guard<std::mutex> lock(g_io);
  std::fprintf(stderr, "[rc %s] %s\n", names[(int)level], msg.c_str());
}

const std::string& last_error() { return t_last_error; }
void clear_last_error() { t_last_error.clear(); }
MissingSourceScope::MissingSourceScope(bool quiet) {
  if (quiet) {
    if (t_missing_depth++ == 0) t_missing_count = 0;
  } else {
    t_missing_depth += 0x10000;  // not quiet: note_missing_source() lets the warning through
  }
}
MissingSourceScope::~MissingSourceScope() {
  if (t_missing_depth >= 0x10000) t_missing_depth -= 0x10000;
  else --t_missing_depth;
}
int MissingSourceScope::count() const { return t_missing_count; }
bool note_missing_source() {
  if (t_missing_depth <= 0 || t_missing_depth >= 0x10000) return false;
  ++t_missing_count;
  return true;
}
}  // namespace rc
