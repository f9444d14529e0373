// The environment switches of the library -- ALL of them: nothing else in dcora_amd/csrc reads the environment.  Each is
// read once per process, except DCORA_PRECOND and DCORA_SOLVER, which are read whenever a problem is created (the
// differential tests build the same problem on both paths in one process).  They force one of two LIVE paths (for differential tests and A/B measurements) or set an
// operational limit; superseded kernel forms are not kept behind switches (DESIGN.md lists what was measured and
// dropped).  INTEGRATION.md section 5 documents them for users.
#pragma once

namespace dcora {
namespace env {

bool init_timing();          // DCORA_INIT_TIMING: laps of the set-up paths on stderr
int precond_mode();          // DCORA_PRECOND=dense|sparse -> 1 | 2 (0: chosen by size, kDensePrecondMaxK)
bool generic_solver();       // DCORA_SOLVER=generic: the thread-per-variable path also where the fused kernels apply
int solver_tcg();            // DCORA_SOLVER_TCG=launch|run -> -1 | +1: the dense tCG run as launches per iteration / as ONE launch (0: by size)
int solver_bc();             // DCORA_SOLVER_BC=pc|split -> +1 | -1: one-launch / three-launch form of the dense tCG step
bool factor_on_host();       // DCORA_FACTOR=host: sparse Cholesky of the preconditioner on host threads
bool fill_on_host();         // DCORA_SP_FILL=host: stored weights formed by host threads and streamed in chunks
bool lanczos_sync();         // DCORA_LANCZOS=sync: one host round trip per Lanczos step (the form used across ranks)
int host_threads();          // DCORA_HOST_THREADS (0: from the affinity mask and the cgroup quota)
double precond_cache_mb();   // DCORA_PRECOND_CACHE_MB (default 8192)
const char *exchange();      // DCORA_EXCHANGE=ipc|staged, or null
const char *exchange_wait(); // DCORA_EXCHANGE_WAIT=device|host, or null
double exchange_timeout_s(); // DCORA_EXCHANGE_TIMEOUT_S (default 120)

}  // namespace env
}  // namespace dcora
