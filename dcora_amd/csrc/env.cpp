#include "env.h"

#include <cstdlib>
#include <cstring>

namespace dcora {
namespace env {
namespace {
bool is(const char *name, const char *value) {
  const char *e = std::getenv(name);
  return e && std::strcmp(e, value) == 0;
}
}  // namespace

bool init_timing() {
  static const bool v = std::getenv("DCORA_INIT_TIMING") != nullptr;
  return v;
}
int precond_mode() {
  return is("DCORA_PRECOND", "dense") ? 1 : is("DCORA_PRECOND", "sparse") ? 2 : 0;
}
bool generic_solver() {
  return is("DCORA_SOLVER", "generic");
}
int solver_tcg() {
  return is("DCORA_SOLVER_TCG", "launch") ? -1 : is("DCORA_SOLVER_TCG", "run") ? 1 : 0;
}
int solver_bc() {
  static const int v = is("DCORA_SOLVER_BC", "pc") ? 1 : is("DCORA_SOLVER_BC", "split") ? -1 : 0;
  return v;
}
bool factor_on_host() {
  static const bool v = is("DCORA_FACTOR", "host");
  return v;
}
bool fill_on_host() {
  static const bool v = is("DCORA_SP_FILL", "host");
  return v;
}
bool lanczos_sync() {
  static const bool v = is("DCORA_LANCZOS", "sync");
  return v;
}
int host_threads() {
  static const int v = [] {
    const char *e = std::getenv("DCORA_HOST_THREADS");
    return e ? (std::atoi(e) > 0 ? std::atoi(e) : 1) : 0;
  }();
  return v;
}
double precond_cache_mb() {
  static const double v = [] {
    const char *e = std::getenv("DCORA_PRECOND_CACHE_MB");
    return e ? std::atof(e) : 8192.0;
  }();
  return v;
}
const char *exchange() {
  static const char *v = std::getenv("DCORA_EXCHANGE");
  return v;
}
const char *exchange_wait() {
  static const char *v = std::getenv("DCORA_EXCHANGE_WAIT");
  return v;
}
double exchange_timeout_s() {
  static const double v = [] {
    const char *e = std::getenv("DCORA_EXCHANGE_TIMEOUT_S");
    const double x = e ? std::atof(e) : 0.0;
    return x > 0 ? x : 120.0;
  }();
  return v;
}

}  // namespace env
}  // namespace dcora
