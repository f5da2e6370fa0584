// A small stand-in for <gtest/gtest.h> (the reference vendors googletest as a git submodule that is
// not checked out under /root/reference/third_party): just the surface the reference's own
// tests/*.cc use - TEST / TEST_F / TEST_P, INSTANTIATE_TEST_SUITE_P with ::testing::Values and an
// optional name generator, EXPECT_* / ASSERT_* with streamed messages, FAIL, GTEST_SKIP,
// InitGoogleTest (--gtest_filter, --gtest_list_tests), RUN_ALL_TESTS - so that those files compile
// UNMODIFIED against the Level-1 library (oracle/Makefile `reftests`) and run on the MI355X.
// Output format follows googletest's ([ RUN ], [ OK ], [ FAILED ], [ SKIPPED ], summary).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <functional>
#include <iostream>
#include <limits>
#include <memory>
#include <sstream>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

namespace testing {

class Message {
public:
  Message() = default;
  Message(const Message& other) { ss_ << other.str(); }
  template <typename T>
  Message& operator<<(const T& v) {
    ss_ << v;
    return *this;
  }
  Message& operator<<(std::ostream& (*manip)(std::ostream&)) {
    ss_ << manip;
    return *this;
  }
  std::string str() const { return ss_.str(); }

private:
  std::ostringstream ss_;
};

namespace internal {

struct State {
  bool failed = false;
  bool skipped = false;
  bool fatal = false;
};
inline State& state() {
  static State s;
  return s;
}

class AssertHelper {
public:
  AssertHelper(const char* file, int line, const std::string& what)
      : file_(file), line_(line), what_(what) {}
  void operator=(const Message& m) const {
    state().failed = true;
    std::cout << file_ << ":" << line_ << ": Failure\n" << what_;
    const std::string extra = m.str();
    if (!extra.empty()) std::cout << "\n" << extra;
    std::cout << std::endl;
  }

private:
  const char* file_;
  int line_;
  std::string what_;
};

class SkipHelper {
public:
  void operator=(const Message& m) const {
    state().skipped = true;
    const std::string extra = m.str();
    if (!extra.empty()) std::cout << "Skipped: " << extra << std::endl;
  }
};

template <typename T, typename = void>
struct Printable : std::false_type {};
template <typename T>
struct Printable<T, decltype(void(std::declval<std::ostream&>() << std::declval<const T&>()))>
    : std::true_type {};

template <typename T>
std::string print(const T& v) {
  if constexpr (std::is_same<T, bool>::value) {
    return v ? "true" : "false";
  } else if constexpr (std::is_enum<T>::value) {
    return std::to_string(static_cast<long long>(v));
  } else if constexpr (std::is_same<T, signed char>::value || std::is_same<T, unsigned char>::value ||
                       std::is_same<T, char>::value) {
    return std::to_string(static_cast<int>(v));
  } else if constexpr (Printable<T>::value) {
    std::ostringstream ss;
    ss.precision(9);
    ss << v;
    return ss.str();
  } else {
    return "<" + std::to_string(sizeof(T)) + "-byte object>";
  }
}

template <typename A, typename B>
std::string cmp_text(const char* op, const char* ea, const char* eb, const A& a, const B& b) {
  return std::string("Expected: (") + ea + ") " + op + " (" + eb + "), actual: " + print(a) +
         " vs " + print(b);
}

// googletest's AlmostEquals: within 4 units in the last place.
inline bool float_almost_equal(float a, float b) {
  if (std::isnan(a) || std::isnan(b)) return false;
  if (a == b) return true;
  auto biased = [](float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? (~u + 1) : (u | 0x80000000u);
  };
  const uint32_t x = biased(a), y = biased(b);
  return (x > y ? x - y : y - x) <= 4;
}

struct TestInfo {
  std::string suite, name;
  std::function<void()> run;  // constructs, SetUp, TestBody, TearDown
};
inline std::vector<TestInfo>& registry() {
  static std::vector<TestInfo> r;
  return r;
}
// parameterised tests are expanded at RUN_ALL_TESTS time
inline std::vector<std::function<void()>>& expanders() {
  static std::vector<std::function<void()>> e;
  return e;
}

inline bool wildcard(const char* pat, const char* str) {
  if (*pat == '\0') return *str == '\0';
  if (*pat == '*') return wildcard(pat + 1, str) || (*str != '\0' && wildcard(pat, str + 1));
  if (*str == '\0') return false;
  return (*pat == '?' || *pat == *str) && wildcard(pat + 1, str + 1);
}
inline bool matches_any(const std::string& patterns, const std::string& name) {
  size_t begin = 0;
  while (begin <= patterns.size()) {
    size_t end = patterns.find(':', begin);
    if (end == std::string::npos) end = patterns.size();
    if (wildcard(patterns.substr(begin, end - begin).c_str(), name.c_str())) return true;
    begin = end + 1;
  }
  return false;
}
inline std::string& filter() {
  static std::string f = "*";
  return f;
}
inline bool& list_only() {
  static bool l = false;
  return l;
}

}  // namespace internal

class Test {
public:
  virtual ~Test() = default;
  virtual void SetUp() {}
  virtual void TearDown() {}
  virtual void TestBody() = 0;
};

template <typename T>
class WithParamInterface {
public:
  using ParamType = T;
  virtual ~WithParamInterface() = default;
  static const T& GetParam() { return *current(); }
  static const T*& current() {
    static const T* p = nullptr;
    return p;
  }
};

template <typename T>
class TestWithParam : public Test, public WithParamInterface<T> {};

template <typename T>
struct TestParamInfo {
  TestParamInfo(const T& p, size_t i) : param(p), index(i) {}
  T param;
  size_t index;
};

namespace internal {
template <typename... Ts>
struct ValueList {
  std::tuple<Ts...> values;
  template <typename P, size_t... I>
  std::vector<P> to(std::index_sequence<I...>) const {
    return std::vector<P>{static_cast<P>(std::get<I>(values))...};
  }
  template <typename P>
  std::vector<P> expand() const {
    return to<P>(std::index_sequence_for<Ts...>{});
  }
};
template <typename C>
struct ValuesInList {
  C values;
  template <typename P>
  std::vector<P> expand() const {
    return std::vector<P>(values.begin(), values.end());
  }
};

template <typename Suite>
struct ParamRegistry {
  using P = typename Suite::ParamType;
  struct Pattern {
    std::string suite, name;
    std::function<Test*()> factory;
  };
  struct Instantiation {
    std::string prefix;
    std::function<std::vector<P>()> params;
    std::function<std::string(const TestParamInfo<P>&)> namer;
  };
  static std::vector<Pattern>& patterns() {
    static std::vector<Pattern> p;
    return p;
  }
  static std::vector<Instantiation>& instantiations() {
    static std::vector<Instantiation> i;
    return i;
  }
  static bool& expander_registered() {
    static bool r = false;
    return r;
  }
  static void ensure_expander() {
    if (expander_registered()) return;
    expander_registered() = true;
    expanders().push_back([] {
      for (const auto& inst : instantiations()) {
        auto params = std::make_shared<std::vector<P>>(inst.params());
        for (const auto& pat : patterns()) {
          for (size_t i = 0; i < params->size(); ++i) {
            const std::string pname = inst.namer ? inst.namer(TestParamInfo<P>((*params)[i], i))
                                                 : std::to_string(i);
            TestInfo info;
            info.suite = inst.prefix + "/" + pat.suite;
            info.name = pat.name + "/" + pname;
            auto factory = pat.factory;
            info.run = [factory, params, i] {
              WithParamInterface<P>::current() = &(*params)[i];
              std::unique_ptr<Test> t(factory());
              t->SetUp();
              if (!state().fatal && !state().skipped) t->TestBody();
              t->TearDown();
              WithParamInterface<P>::current() = nullptr;
            };
            registry().push_back(std::move(info));
          }
        }
      }
    });
  }
  static int add_pattern(const char* suite, const char* name, std::function<Test*()> factory) {
    ensure_expander();
    patterns().push_back(Pattern{suite, name, std::move(factory)});
    return 0;
  }
  template <typename Gen>
  static int add_instantiation(const char* prefix, Gen gen,
                               std::function<std::string(const TestParamInfo<P>&)> namer = nullptr) {
    ensure_expander();
    instantiations().push_back(
        Instantiation{prefix, [gen] { return gen.template expand<P>(); }, std::move(namer)});
    return 0;
  }
};

inline int add_test(const char* suite, const char* name, std::function<Test*()> factory) {
  TestInfo info;
  info.suite = suite;
  info.name = name;
  info.run = [factory] {
    std::unique_ptr<Test> t(factory());
    t->SetUp();
    if (!state().fatal && !state().skipped) t->TestBody();
    t->TearDown();
  };
  registry().push_back(std::move(info));
  return 0;
}
}  // namespace internal

template <typename... Ts>
internal::ValueList<Ts...> Values(Ts... vs) {
  return internal::ValueList<Ts...>{std::make_tuple(vs...)};
}
template <typename C>
internal::ValuesInList<C> ValuesIn(const C& c) {
  return internal::ValuesInList<C>{c};
}

inline void InitGoogleTest(int* argc, char** argv) {
  int out = 1;
  for (int i = 1; i < *argc; ++i) {
    const std::string a = argv[i];
    if (a.rfind("--gtest_filter=", 0) == 0)
      internal::filter() = a.substr(15);
    else if (a == "--gtest_list_tests")
      internal::list_only() = true;
    else if (a.rfind("--gtest_", 0) == 0)
      ;  // other googletest flags are accepted and ignored
    else
      argv[out++] = argv[i];
  }
  *argc = out;
}

inline int RunAllTests() {
  for (auto& expand : internal::expanders()) expand();
  std::string positive = internal::filter(), negative;
  const size_t dash = positive.find('-');
  if (dash != std::string::npos) {
    negative = positive.substr(dash + 1);
    positive = positive.substr(0, dash);
  }
  if (positive.empty()) positive = "*";
  int ran = 0, passed = 0, skipped = 0;
  std::vector<std::string> failed;
  for (auto& t : internal::registry()) {
    const std::string full = t.suite + "." + t.name;
    if (!internal::matches_any(positive, full)) continue;
    if (!negative.empty() && internal::matches_any(negative, full)) continue;
    if (internal::list_only()) {
      std::cout << full << "\n";
      continue;
    }
    internal::state() = internal::State();
    std::cout << "[ RUN      ] " << full << std::endl;
    try {
      t.run();
    } catch (const std::exception& e) {
      internal::state().failed = true;
      std::cout << "unknown file: Failure\nC++ exception with description \"" << e.what()
                << "\" thrown in the test body." << std::endl;
    } catch (...) {
      internal::state().failed = true;
      std::cout << "unknown file: Failure\nUnknown C++ exception thrown in the test body."
                << std::endl;
    }
    ++ran;
    if (internal::state().failed) {
      failed.push_back(full);
      std::cout << "[  FAILED  ] " << full << std::endl;
    } else if (internal::state().skipped) {
      ++skipped;
      std::cout << "[  SKIPPED ] " << full << std::endl;
    } else {
      ++passed;
      std::cout << "[       OK ] " << full << std::endl;
    }
  }
  if (internal::list_only()) return 0;
  std::cout << "[==========] " << ran << " tests ran.\n[  PASSED  ] " << passed << " tests.\n";
  if (skipped) std::cout << "[  SKIPPED ] " << skipped << " tests.\n";
  if (!failed.empty()) {
    std::cout << "[  FAILED  ] " << failed.size() << " tests, listed below:\n";
    for (const auto& f : failed) std::cout << "[  FAILED  ] " << f << "\n";
  }
  std::cout << std::flush;
  return failed.empty() ? 0 : 1;
}

}  // namespace testing

#define RUN_ALL_TESTS() ::testing::RunAllTests()

#define GTEST_CLASS_(suite, name) suite##_##name##_Test

#define TEST(suite, name)                                                              \
  class GTEST_CLASS_(suite, name) : public ::testing::Test {                           \
    void TestBody() override;                                                          \
    static int registered_;                                                            \
  };                                                                                   \
  int GTEST_CLASS_(suite, name)::registered_ = ::testing::internal::add_test(          \
      #suite, #name, [] { return static_cast<::testing::Test*>(new GTEST_CLASS_(suite, name)); }); \
  void GTEST_CLASS_(suite, name)::TestBody()

#define TEST_F(fixture, name)                                                          \
  class GTEST_CLASS_(fixture, name) : public fixture {                                 \
    void TestBody() override;                                                          \
    static int registered_;                                                            \
  };                                                                                   \
  int GTEST_CLASS_(fixture, name)::registered_ = ::testing::internal::add_test(        \
      #fixture, #name,                                                                 \
      [] { return static_cast<::testing::Test*>(new GTEST_CLASS_(fixture, name)); });  \
  void GTEST_CLASS_(fixture, name)::TestBody()

#define TEST_P(suite, name)                                                            \
  class GTEST_CLASS_(suite, name) : public suite {                                     \
    void TestBody() override;                                                          \
    static int registered_;                                                            \
  };                                                                                   \
  int GTEST_CLASS_(suite, name)::registered_ =                                         \
      ::testing::internal::ParamRegistry<suite>::add_pattern(                          \
          #suite, #name,                                                               \
          [] { return static_cast<::testing::Test*>(new GTEST_CLASS_(suite, name)); }); \
  void GTEST_CLASS_(suite, name)::TestBody()

#define INSTANTIATE_TEST_SUITE_P(prefix, suite, ...)                                   \
  static int gtest_inst_##prefix##_##suite =                                           \
      ::testing::internal::ParamRegistry<suite>::add_instantiation(#prefix, __VA_ARGS__)
#define INSTANTIATE_TEST_CASE_P INSTANTIATE_TEST_SUITE_P

#define GTEST_MESSAGE_AT_(what) \
  ::testing::internal::AssertHelper(__FILE__, __LINE__, what) = ::testing::Message()
#define GTEST_NONFATAL_(what) GTEST_MESSAGE_AT_(what)
#define GTEST_FATAL_(what) \
  return ::testing::internal::state().fatal = true, GTEST_MESSAGE_AT_(what)

#define FAIL() GTEST_FATAL_("Failed")
#define ADD_FAILURE() GTEST_NONFATAL_("Failed")
#define SUCCEED() ::testing::Message()
#define GTEST_SKIP() return ::testing::internal::SkipHelper() = ::testing::Message()

#define GTEST_CMP_(fail, op, opname, a, b)                                            \
  if (const auto& gtest_a_ = (a); true)                                                \
    if (const auto& gtest_b_ = (b); gtest_a_ op gtest_b_)                              \
      ;                                                                                \
    else                                                                               \
      fail(::testing::internal::cmp_text(opname, #a, #b, gtest_a_, gtest_b_))

#define EXPECT_EQ(a, b) GTEST_CMP_(GTEST_NONFATAL_, ==, "==", a, b)
#define EXPECT_NE(a, b) GTEST_CMP_(GTEST_NONFATAL_, !=, "!=", a, b)
#define EXPECT_LT(a, b) GTEST_CMP_(GTEST_NONFATAL_, <, "<", a, b)
#define EXPECT_LE(a, b) GTEST_CMP_(GTEST_NONFATAL_, <=, "<=", a, b)
#define EXPECT_GT(a, b) GTEST_CMP_(GTEST_NONFATAL_, >, ">", a, b)
#define EXPECT_GE(a, b) GTEST_CMP_(GTEST_NONFATAL_, >=, ">=", a, b)
#define ASSERT_EQ(a, b) GTEST_CMP_(GTEST_FATAL_, ==, "==", a, b)
#define ASSERT_NE(a, b) GTEST_CMP_(GTEST_FATAL_, !=, "!=", a, b)
#define ASSERT_LT(a, b) GTEST_CMP_(GTEST_FATAL_, <, "<", a, b)
#define ASSERT_LE(a, b) GTEST_CMP_(GTEST_FATAL_, <=, "<=", a, b)
#define ASSERT_GT(a, b) GTEST_CMP_(GTEST_FATAL_, >, ">", a, b)
#define ASSERT_GE(a, b) GTEST_CMP_(GTEST_FATAL_, >=, ">=", a, b)

#define GTEST_BOOL_(fail, expr, want)                                                  \
  if (static_cast<bool>(expr) == want)                                                 \
    ;                                                                                  \
  else                                                                                 \
    fail(std::string("Value of: " #expr "\n  Actual: ") + (want ? "false" : "true") +  \
         "\nExpected: " + (want ? "true" : "false"))
#define EXPECT_TRUE(e) GTEST_BOOL_(GTEST_NONFATAL_, e, true)
#define EXPECT_FALSE(e) GTEST_BOOL_(GTEST_NONFATAL_, e, false)
#define ASSERT_TRUE(e) GTEST_BOOL_(GTEST_FATAL_, e, true)
#define ASSERT_FALSE(e) GTEST_BOOL_(GTEST_FATAL_, e, false)

#define GTEST_NEAR_(fail, a, b, tol)                                                   \
  if (const double gtest_d_ = std::fabs(static_cast<double>(a) - static_cast<double>(b)); \
      gtest_d_ <= static_cast<double>(tol))                                            \
    ;                                                                                  \
  else                                                                                 \
    fail(std::string("The difference between " #a " and " #b " is ") +                 \
         ::testing::internal::print(gtest_d_) + ", which exceeds " #tol ", where\n" #a \
         " evaluates to " + ::testing::internal::print(a) + ",\n" #b " evaluates to " + \
         ::testing::internal::print(b))
#define EXPECT_NEAR(a, b, tol) GTEST_NEAR_(GTEST_NONFATAL_, a, b, tol)
#define ASSERT_NEAR(a, b, tol) GTEST_NEAR_(GTEST_FATAL_, a, b, tol)

#define GTEST_FLOAT_EQ_(fail, a, b)                                                    \
  if (::testing::internal::float_almost_equal(static_cast<float>(a), static_cast<float>(b))) \
    ;                                                                                  \
  else                                                                                 \
    fail(::testing::internal::cmp_text("~=", #a, #b, static_cast<float>(a), static_cast<float>(b)))
#define EXPECT_FLOAT_EQ(a, b) GTEST_FLOAT_EQ_(GTEST_NONFATAL_, a, b)
#define ASSERT_FLOAT_EQ(a, b) GTEST_FLOAT_EQ_(GTEST_FATAL_, a, b)

#define GTEST_THROW_(fail, stmt, ex)                                                   \
  if (bool gtest_caught_ = false; true) {                                              \
    try {                                                                              \
      stmt;                                                                            \
    } catch (const ex&) {                                                              \
      gtest_caught_ = true;                                                            \
    } catch (...) {                                                                    \
    }                                                                                  \
    if (!gtest_caught_) goto GTEST_CONCAT_(gtest_throw_fail_, __LINE__);               \
  } else                                                                               \
    GTEST_CONCAT_(gtest_throw_fail_, __LINE__)                                         \
        : fail("Expected: " #stmt " throws an exception of type " #ex ".")
#define GTEST_CONCAT_I_(a, b) a##b
#define GTEST_CONCAT_(a, b) GTEST_CONCAT_I_(a, b)
#define EXPECT_THROW(stmt, ex) GTEST_THROW_(GTEST_NONFATAL_, stmt, ex)
#define ASSERT_THROW(stmt, ex) GTEST_THROW_(GTEST_FATAL_, stmt, ex)
#define EXPECT_NO_THROW(stmt)                                                          \
  try {                                                                                \
    stmt;                                                                              \
  } catch (...) {                                                                      \
    GTEST_NONFATAL_("Expected: " #stmt " doesn't throw an exception.");                \
  }
