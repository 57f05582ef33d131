// Core enums, Status and the logging/check shim of the host layer.
// Mirrors kuiper/include/base/base.h:11-154 (DeviceType, DataType, StatusCode, Status,
// STATUS_CHECK, model::ModelBufferType) with one addition: DeviceType::kDeviceHIP.
// glog is not a dependency here; LOG/CHECK below keep the observable convention of the
// reference (a failed CHECK or LOG(FATAL) prints the message and aborts).
#ifndef KX_HOST_BASE_BASE_H_
#define KX_HOST_BASE_BASE_H_
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>

#define UNUSED(expr) \
  do {               \
    (void)(expr);    \
  } while (0)

namespace model {
// kuiper/include/base/base.h:12-32
enum class ModelBufferType {
  kInputTokens = 0,
  kInputEmbeddings = 1,
  kOutputRMSNorm = 2,
  kKeyCache = 3,
  kValueCache = 4,
  kQuery = 5,
  kInputPos = 6,
  kScoreStorage = 7,
  kOutputMHA = 8,
  kAttnOutput = 9,
  kW1Output = 10,
  kW2Output = 11,
  kW3Output = 12,
  kFFNRMSNorm = 13,
  kForwardOutput = 15,
  kForwardOutputCPU = 16,
  kSinCache = 17,
  kCosCache = 18,
};
}  // namespace model

namespace base {
// kuiper/include/base/base.h:36-40; kDeviceHIP is the new member the drop-in adds.
enum class DeviceType : uint8_t {
  kDeviceUnknown = 0,
  kDeviceCPU = 1,
  kDeviceCUDA = 2,  // numbering kept; no kernels are registered for it in this library
  kDeviceHIP = 3,
};

enum class DataType : uint8_t {
  kDataTypeUnknown = 0,
  kDataTypeFp32 = 1,
  kDataTypeInt8 = 2,
  kDataTypeInt32 = 3,
  kDataTypeFp16 = 4,  // IEEE half; KV cache rows only (not a reference dtype, base.h:25-30)
};

inline size_t DataTypeSize(DataType t) {
  switch (t) {
    case DataType::kDataTypeFp32: return 4;
    case DataType::kDataTypeInt8: return 1;
    case DataType::kDataTypeInt32: return 4;
    case DataType::kDataTypeFp16: return 2;
    default: return 0;
  }
}

class NoCopyable {
 protected:
  NoCopyable() = default;
  ~NoCopyable() = default;
  NoCopyable(const NoCopyable&) = delete;
  NoCopyable& operator=(const NoCopyable&) = delete;
};

enum StatusCode : uint8_t {
  kSuccess = 0,
  kFunctionUnImplement = 1,
  kPathNotValid = 2,
  kModelParseError = 3,
  kInternalError = 5,
  kKeyValueHasExist = 6,
  kInvalidArgument = 7,
};

class Status {
 public:
  Status(int code = StatusCode::kSuccess, std::string err_message = "")
      : code_(code), message_(std::move(err_message)) {}
  Status& operator=(int code) {
    code_ = code;
    return *this;
  }
  bool operator==(int code) const { return code_ == code; }
  bool operator!=(int code) const { return code_ != code; }
  operator int() const { return code_; }
  operator bool() const { return code_ == kSuccess; }  // base.cpp:31-33
  int32_t get_err_code() const { return code_; }
  const std::string& get_err_msg() const { return message_; }
  void set_err_msg(const std::string& m) { message_ = m; }

 private:
  int code_ = StatusCode::kSuccess;
  std::string message_;
};

namespace error {
inline Status Success(const std::string& m = "") { return Status{kSuccess, m}; }
inline Status FunctionNotImplement(const std::string& m = "") { return Status{kFunctionUnImplement, m}; }
inline Status PathNotValid(const std::string& m = "") { return Status{kPathNotValid, m}; }
inline Status ModelParseError(const std::string& m = "") { return Status{kModelParseError, m}; }
inline Status InternalError(const std::string& m = "") { return Status{kInternalError, m}; }
inline Status KeyHasExits(const std::string& m = "") { return Status{kKeyValueHasExist, m}; }
inline Status InvalidArgument(const std::string& m = "") { return Status{kInvalidArgument, m}; }
}  // namespace error

inline std::ostream& operator<<(std::ostream& os, const Status& x) {
  os << x.get_err_msg();
  return os;
}

// ---- minimal stand-in for the glog calls the reference uses ----
namespace logging {
enum Severity { INFO = 0, WARNING = 1, ERROR = 2, FATAL = 3 };
int& min_severity();  // messages below it are dropped (default WARNING)
class Message {
 public:
  Message(const char* file, int line, Severity sev) : sev_(sev) {
    static const char* names[] = {"I", "W", "E", "F"};
    ss_ << names[sev] << " " << file << ":" << line << "] ";
  }
  ~Message() {
    if (sev_ >= min_severity()) std::cerr << ss_.str() << std::endl;
    if (sev_ == FATAL) std::abort();
  }
  std::ostream& stream() { return ss_; }

 private:
  Severity sev_;
  std::ostringstream ss_;
};
struct Voidify {
  void operator&(std::ostream&) {}
};
}  // namespace logging
}  // namespace base

#define LOG(sev) ::base::logging::Message(__FILE__, __LINE__, ::base::logging::sev).stream()
#define LOG_IF(sev, cond) \
  !(cond) ? (void)0 : ::base::logging::Voidify() & LOG(sev)
#define CHECK(cond) \
  (cond) ? (void)0 : ::base::logging::Voidify() & LOG(FATAL) << "Check failed: " #cond " "
#define CHECK_OP_(a, b, op) CHECK((a)op(b)) << "(" << (a) << " vs " << (b) << ") "
#define CHECK_EQ(a, b) CHECK_OP_(a, b, ==)
#define CHECK_NE(a, b) CHECK((a) != (b))
#define CHECK_LT(a, b) CHECK_OP_(a, b, <)
#define CHECK_LE(a, b) CHECK_OP_(a, b, <=)
#define CHECK_GT(a, b) CHECK_OP_(a, b, >)
#define CHECK_GE(a, b) CHECK_OP_(a, b, >=)

// kuiper/include/base/base.h:123-134
#define STATUS_CHECK(call)                                                              \
  do {                                                                                  \
    const base::Status& status_ = call;                                                 \
    if (!status_) {                                                                     \
      LOG(FATAL) << "Infer error\n File:" << __FILE__ << " Line:" << __LINE__           \
                 << "\n Error code:" << int(status_) << "\n Error msg:" << status_.get_err_msg(); \
    }                                                                                   \
  } while (0)

#endif  // KX_HOST_BASE_BASE_H_
