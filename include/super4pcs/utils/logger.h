// Drop-in for src/super4pcs/utils/logger.h:55-118 (GlobalRegistration::Utils::Logger): same enum, same
// Log<level>(args...) interface; appears in the matcher constructors (super4pcs.h:62-63).
#ifndef S4P_FACADE_LOGGER_H_
#define S4P_FACADE_LOGGER_H_
#include <iostream>

namespace GlobalRegistration {
namespace Utils {

enum LogLevel { NoLog = 0, ErrorReport = 1, Verbose = 2 };

class Logger {
 public:
  inline Logger(LogLevel loglevel = Verbose) : logLevel_(loglevel) {}
  inline void setLogLevel(LogLevel loglevel) { logLevel_ = loglevel; }
  inline LogLevel logLevel() const { return logLevel_; }

  template <LogLevel level, typename... Args>
  inline void Log(const Args&... args) const {
    if (int(logLevel_) < int(level) || level == NoLog) return;
    std::ostream& os = (level == ErrorReport) ? std::cerr : std::cout;
    emit(os, args...);
    os << std::endl;
  }

 private:
  static inline void emit(std::ostream&) {}
  template <typename First, typename... Rest>
  static inline void emit(std::ostream& os, const First& a, const Rest&... rest) { os << a; emit(os, rest...); }
  LogLevel logLevel_;
};

}  // namespace Utils
}  // namespace GlobalRegistration
#endif
