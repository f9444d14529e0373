// Neighbour exchange of public poses between the ranks of one node (see exchange.h).
#include "exchange.h"
#include "env.h"

#include <fcntl.h>
#include <sched.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>

namespace dcora {

namespace {

using Clock = std::chrono::steady_clock;
inline double since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }
constexpr int kMaxDst = 8;
struct PostDst {
  double *base[kMaxDst];
  int n;
};

// getSharedStateDicts of one agent, written where its neighbours read it: element i of the packed r x (d+1)count
// block goes to every destination slot (the peers' halo buffers over xGMI, or the shared host segment); the
// workgroup that finishes last publishes the sequence number.  Every storing wave drains its stores and the
// workgroup's lane 0 releases at system scope before it arrives (MI355X_MICROARCH.md, valid producer forms).
__global__ __launch_bounds__(kBlock) void k_post_public(int r, int ncols, const int *__restrict__ src,
                                                        const double *__restrict__ X, PostDst dst, PostDst dflag,
                                                        unsigned *arrive, volatile uint64_t *flag, uint64_t seq) {
  const long N = (long)ncols * r;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < N; i += (long)gridDim.x * kBlock) {
    const int c = (int)(i / r), t = (int)(i - (long)c * r);
    const double v = X[(size_t)src[c] * r + t];
    for (int q = 0; q < dst.n; ++q) dst.base[q][i] = v;
  }
  __threadfence_system();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = atomicAdd(arrive, 1u);
    if (prev == gridDim.x - 1) {
      __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // the flag word next to each destination's slot (what that rank's scatter kernel polls), then the host-visible one
      for (int q = 0; q < dflag.n; ++q)
        __hip_atomic_store((uint64_t *)dflag.base[q], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store((uint64_t *)flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// updateNeighborStates on the consumer's side, with the wait inside: lane 0 of every workgroup polls the flag word
// (system-scope relaxed loads, s_sleep between polls; bounded -- a producer that never arrives raises `failed` and the
// kernel ends), one system-scope acquire, then the slot is read with system-scope loads and scattered into the mirror of
// X.  The workgroup that finishes last tells the producer that the slot has been read (back-pressure of post()).
constexpr long long kWaitBudgetTicks = 20LL * 100000000LL;  // 20 s of the 100 MHz wall clock
__global__ __launch_bounds__(kBlock) void k_wait_scatter(int r, int ncols, const int *__restrict__ cols,
                                                         const double *src, double *__restrict__ X,
                                                         const uint64_t *flag, uint64_t want, unsigned *arrive,
                                                         uint64_t *consumed, uint32_t *failed) {
  __shared__ int s_bad;
  if (threadIdx.x == 0) {
    int bad = 0;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
      __builtin_amdgcn_s_sleep(16);
      if (wall_clock64() - t0 > kWaitBudgetTicks ||
          __hip_atomic_load(failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) {
        bad = 1;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_bad = bad;
  }
  __syncthreads();
  if (s_bad) {
    if (threadIdx.x == 0) __hip_atomic_store(failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  const long N = (long)ncols * r;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < N; i += (long)gridDim.x * kBlock) {
    const int c = (int)(i / r), t = (int)(i - (long)c * r);
    X[(size_t)cols[c] * r + t] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = atomicAdd(arrive, 1u);
    if (prev == gridDim.x - 1) {
      __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(consumed, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// the hosted agents' two evaluation scalars into the shared segment: the device store is the all-gather
__global__ void k_eval_publish(int nh, const int *__restrict__ hosted, const double *__restrict__ ev, ShmEval *slots,
                               uint64_t seq) {
  const int i = threadIdx.x;
  if (i < nh) {
    const int a = hosted[i];
    ShmEval *e = slots + a;
    __hip_atomic_store((double *)&e->g2, ev[2 * a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store((double *)&e->xeg, ev[2 * a + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store((uint64_t *)&e->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ void k_selftest_write(double *dst, int count, double base) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) dst[i] = base + i;
  __threadfence_system();
}

// ---- link check (Exchange::link_check): the producer / consumer forms of k_post_public / k_wait_scatter on a 4 KB
//      probe, one workgroup each ----
__global__ __launch_bounds__(256) void k_probe_post(double *dst, uint64_t *dflag, volatile uint64_t *hflag, double base,
                                                    uint64_t seq) {
  for (int i = threadIdx.x; i < kProbeDoubles; i += 256) dst[i] = base + i;
  __threadfence_system();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (dflag) __hip_atomic_store(dflag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store((uint64_t *)hflag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// result: 1 = flag seen and every word equal, 2 = the flag never came (budget), 3 = words differ
__global__ __launch_bounds__(256) void k_probe_wait(const double *src, const uint64_t *flag, uint64_t want,
                                                    long long budget_ticks, int poll, double base, int *result) {
  __shared__ int s_bad, s_diff;
  if (threadIdx.x == 0) {
    int bad = 0;
    s_diff = 0;
    if (poll) {
      const long long t0 = wall_clock64();
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
        __builtin_amdgcn_s_sleep(16);
        if (wall_clock64() - t0 > budget_ticks) {
          bad = 1;
          break;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_bad = bad;
  }
  __syncthreads();
  if (s_bad) {
    if (threadIdx.x == 0) __hip_atomic_store(result, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  int diff = 0;
  for (int i = threadIdx.x; i < kProbeDoubles; i += 256)
    diff += __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != base + i;
  if (diff) atomicAdd(&s_diff, diff);
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(result, s_diff ? 3 : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// start time of a process in clock ticks since boot (field 22 of /proc/<pid>/stat), 0 when it cannot be read
uint64_t proc_start_time(int pid) {
  char path[64], buf[1024];
  std::snprintf(path, sizeof path, "/proc/%d/stat", pid);
  FILE *fp = std::fopen(path, "r");
  if (!fp) return 0;
  const size_t n = std::fread(buf, 1, sizeof buf - 1, fp);
  std::fclose(fp);
  buf[n] = 0;
  const char *p = std::strrchr(buf, ')');  // the command may hold spaces and parentheses: fields resume behind the last ')'
  if (!p) return 0;
  int field = 2;
  for (++p; *p; ++p)
    if (*p == ' ' && ++field == 22) return std::strtoull(p + 1, nullptr, 10);
  return 0;
}
// is the creator of a segment still the process that wrote the header?
bool creator_alive(const ShmHeader *h) {
  if (h->creator_pid <= 0) return false;
  if (!(kill((pid_t)h->creator_pid, 0) == 0 || errno == EPERM)) return false;
  const uint64_t now = proc_start_time(h->creator_pid);
  return now == 0 || h->creator_start == 0 || now == h->creator_start;  // (no /proc: the pid test alone)
}

// host polling with back-off: a burst of pause instructions, then the core is handed over between polls (a rank per
// core is not guaranteed: the four-ranks-on-one-GPU rehearsal runs on whatever cores the container has)
inline void polite_spin(unsigned &spins) {
  ++spins;
  if (spins < 2048u) {
    __builtin_ia32_pause();
  } else if (spins < 8192u) {
    sched_yield();
  } else {
    usleep(50);
  }
}

}  // namespace

double exchange_timeout_s() { return env::exchange_timeout_s(); }

std::atomic<int> g_probe_fault_rounds{0};

int Exchange::usage(const std::string &msg, int code) {
  set_last_error("exchange (rank " + std::to_string(rank) + "): " + msg);
  return code;
}

int Exchange::fail(const std::string &msg, int code) {
  set_last_error("exchange (rank " + std::to_string(rank) + "): " + msg);
  if (hdr_) hdr_->failed.store(1);
  return code;
}

int Exchange::num_peers() const {
  std::set<int> p;
  for (const auto &v : dests_) p.insert(v.begin(), v.end());
  return (int)p.size();
}

Exchange::~Exchange() {
  if (s_) (void)hipSetDevice(s_->x_device());
  for (int q = 0; q < kMaxRanks; ++q)
    if (opened_[q] && peer_halo_[q]) (void)hipIpcCloseMemHandle(peer_halo_[q]);
  if (registered_ && map_) (void)hipHostUnregister(map_);
  if (map_) munmap(map_, map_bytes_);
  if (rank == 0 && !name_.empty()) shm_unlink(name_.c_str());
}

// Rank 0 removes whatever carries the name, creates the segment and publishes the magic word last; the others attach
// once it is there.  A rank that opened the name BEFORE rank 0 replaced it holds a stale segment (a crashed job, a reused
// job name): it recognises that by the creator's process being gone or by the name resolving to another inode by now,
// drops the mapping and opens the name again.  Two LIVE jobs under one name on one node remain a caller's error.
int Exchange::open_segment(const char *job_name, size_t bytes) {
  name_ = std::string("/dcora_") + job_name;
  const auto t0 = Clock::now();
  if (rank == 0) {
    shm_unlink(name_.c_str());
    const int fd = shm_open(name_.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return fail("shm_open(create " + name_ + ") failed: " + std::strerror(errno), DCORA_ERR_IO);
    if (ftruncate(fd, (off_t)bytes) != 0) {
      close(fd);
      return fail("ftruncate failed: " + std::string(std::strerror(errno)), DCORA_ERR_IO);
    }
    map_ = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (map_ == MAP_FAILED) {
      map_ = nullptr;
      return fail("mmap failed: " + std::string(std::strerror(errno)), DCORA_ERR_IO);
    }
    map_bytes_ = bytes;
    hdr_ = (ShmHeader *)map_;
    std::memset(map_, 0, bytes);
    hdr_->world = (uint32_t)world;
    hdr_->R = (uint32_t)R_;
    hdr_->slot_doubles = slot_;
    hdr_->total_bytes = bytes;
    hdr_->creator_pid = (int32_t)getpid();
    hdr_->creator_start = proc_start_time((int)getpid());
    hdr_->magic.store(kShmMagic, std::memory_order_release);
    return DCORA_OK;
  }
  for (;;) {
    if (since(t0) > exchange_timeout_s()) return fail("segment " + name_ + " did not appear", DCORA_ERR_IO);
    const int fd = shm_open(name_.c_str(), O_RDWR, 0600);
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb) != 0 || (size_t)sb.st_size < bytes) {
      // a segment of another size under the name: a live job of another shape says so in its header (a stale one, or
      // one rank 0 is still creating, is waited out)
      if (fd >= 0 && (size_t)sb.st_size >= sizeof(ShmHeader)) {
        void *hp = mmap(nullptr, sizeof(ShmHeader), PROT_READ, MAP_SHARED, fd, 0);
        if (hp != MAP_FAILED) {
          ShmHeader *h = (ShmHeader *)hp;
          const bool live = h->magic.load(std::memory_order_acquire) == kShmMagic && creator_alive(h);
          const bool other = live && (h->world != (uint32_t)world || h->R != (uint32_t)R_ || h->slot_doubles != slot_ ||
                                      h->total_bytes != bytes);
          if (other && since(t0) > 1.0) {  // (a second: a crashed job's segment is replaced by rank 0 within that)
            // a LOCAL error only: the segment belongs to somebody else's live job and is not written to (raising its
            // `failed` word took that healthy job down); this job's other ranks time out on DCORA_EXCHANGE_TIMEOUT_S
            munmap(hp, sizeof(ShmHeader));
            close(fd);
            return fail("segment " + name_ + " belongs to a job of another shape (a live job under the same name?)",
                        DCORA_ERR_BAD_ARG);
          }
          munmap(hp, sizeof(ShmHeader));
        }
      }
      if (fd >= 0) close(fd);
      usleep(2000);
      continue;
    }
    void *mp = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (mp == MAP_FAILED) return fail("mmap failed: " + std::string(std::strerror(errno)), DCORA_ERR_IO);
    ShmHeader *h = (ShmHeader *)mp;
    bool good = false;
    const auto t1 = Clock::now();
    while (since(t1) < 0.25) {  // a fresh segment is initialised within microseconds of its creation
      if (h->magic.load(std::memory_order_acquire) == kShmMagic) {
        good = true;
        break;
      }
      usleep(500);
    }
    if (good) {
      // still the segment the name stands for, and its creator is alive?
      struct stat now;
      const int fd2 = shm_open(name_.c_str(), O_RDONLY, 0600);
      const bool same = fd2 >= 0 && fstat(fd2, &now) == 0 && now.st_ino == sb.st_ino && now.st_dev == sb.st_dev;
      if (fd2 >= 0) close(fd2);
      good = same && creator_alive(h);
    }
    if (!good) {
      munmap(mp, bytes);
      usleep(2000);
      continue;
    }
    if (h->world != (uint32_t)world || h->R != (uint32_t)R_ || h->slot_doubles != slot_ || h->total_bytes != bytes) {
      h->failed.store(1);  // the job's other ranks give up as well instead of waiting for this one
      munmap(mp, bytes);
      return fail("segment " + name_ + " belongs to a job of another shape (a live job under the same name?)",
                  DCORA_ERR_BAD_ARG);
    }
    map_ = mp;
    map_bytes_ = bytes;
    hdr_ = h;
    return DCORA_OK;
  }
}

// shared segment: header | per-rank records | flags [2][R] | evaluation slots [2][R] | staged poses [2][R][slot] | X
int Exchange::map_segment(const char *job_name, size_t x_doubles) {
  const int R = R_;
  size_t off = align_up(sizeof(ShmHeader), 64);
  const size_t off_ranks = off;
  off += sizeof(ShmRank) * world;
  off_flags_ = off;
  off += sizeof(ShmFlag) * 2 * R;
  off_evals_ = off;
  off += sizeof(ShmEval) * 2 * (R + world);  // R agent slots + one heartbeat slot per rank, double-buffered
  off_consumed_ = off;
  off += sizeof(ShmFlag) * (size_t)world * R;
  off_red_ = off;
  off += sizeof(ShmRed) * 2 * (size_t)world;
  off_probe_flags_ = off;  // link check: [reader][writer] flag words, then result words
  off += sizeof(ShmFlag) * (size_t)world * world;
  off_probe_res_ = off;
  off += sizeof(ShmFlag) * (size_t)world * world;
  off = align_up(off, 4096);
  off_probe_stage_ = off;  // ... and 4 KB per (reader, writer) for the staged transport's probe
  off += sizeof(double) * kProbeDoubles * (size_t)world * world;
  off = align_up(off, 4096);
  off_staged_ = off;
  off += sizeof(double) * 2 * R * slot_;
  off = align_up(off, 4096);
  off_x_ = off;
  off += sizeof(double) * x_doubles;
  const size_t total = align_up(off, 4096);
  const int rc = open_segment(job_name, total);
  if (rc) return rc;
  ranks_ = (ShmRank *)((char *)map_ + off_ranks);
  flags_ = (ShmFlag *)((char *)map_ + off_flags_);
  evals_ = (ShmEval *)((char *)map_ + off_evals_);
  consumed_ = (ShmFlag *)((char *)map_ + off_consumed_);
  red_ = (ShmRed *)((char *)map_ + off_red_);
  staged_ = (double *)((char *)map_ + off_staged_);
  xarea_ = (double *)((char *)map_ + off_x_);
  return DCORA_OK;
}

int Exchange::barrier(double timeout_s) {
  if (world == 1) return DCORA_OK;
  timeout_s = std::min(timeout_s, exchange_timeout_s());
  const auto t0 = Clock::now();
  const uint32_t gen = hdr_->bar_gen.load(std::memory_order_acquire);
  if (hdr_->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)world) {
    hdr_->bar_count.store(0, std::memory_order_relaxed);
    hdr_->bar_gen.fetch_add(1, std::memory_order_release);
    return DCORA_OK;
  }
  unsigned spins = 0;
  while (hdr_->bar_gen.load(std::memory_order_acquire) == gen) {
    if ((++spins & 1023u) == 0) {
      if (hdr_->failed.load()) return fail("another rank failed", DCORA_ERR_HIP);
      if (since(t0) > timeout_s) return fail("barrier timed out", DCORA_ERR_HIP);
      sched_yield();
    }
  }
  return DCORA_OK;
}

int Exchange::init(ExchangeSession *s, const char *job_name) {
  s_ = s;
  rank = s->x_rank();
  world = s->x_world();
  R_ = s->x_num_agents();
  const int R = R_;
  if (world > kMaxRanks) return fail("too many ranks", DCORA_ERR_UNSUPPORTED);
  if (!job_name || !*job_name) return fail("empty job name", DCORA_ERR_BAD_ARG);
  DCORA_HIP(hipSetDevice(s->x_device()));
  const int per = (R + world - 1) / world;
  owner_.resize(R);
  size_t maxcols = 1;
  std::vector<int> hosted;
  for (int a = 0; a < R; ++a) {
    const XAgentView v = s->x_agent(a);
    owner_[a] = a / per;
    maxcols = std::max(maxcols, (size_t)v.ncols);
    if (v.hosted) hosted.push_back(a);
    if (v.hosted != (owner_[a] == rank)) return fail("agent-to-rank map out of step with the session", DCORA_ERR_BAD_ARG);
  }
  n_hosted_ = (int)hosted.size();
  slot_ = align_up(maxcols * (size_t)s->x_rank_r(), 16);
  dests_.assign(R, {});
  needed_.assign(R, 0);
  for (int a = 0; a < R; ++a) {
    const XAgentView va = s->x_agent(a);
    for (int q : *va.neighbors) {
      const XAgentView vq = s->x_agent(q);
      if (va.hosted && owner_[q] != rank && std::find(dests_[a].begin(), dests_[a].end(), owner_[q]) == dests_[a].end())
        dests_[a].push_back(owner_[q]);
      if (!va.hosted && vq.hosted) needed_[a] = 1;
    }
  }
  for (int a = 0; a < R; ++a)
    if ((int)dests_[a].size() > kMaxDst) return fail("an agent has neighbours on more than 8 other ranks", DCORA_ERR_UNSUPPORTED);
  seq_.assign(R, 0);

  int rc = map_segment(job_name, (size_t)s->x_rank_r() * (size_t)s->x_num_cols());
  if (rc) return rc;
  {
    const hipError_t e = hipHostRegister(map_, map_bytes_, hipHostRegisterMapped | hipHostRegisterPortable);
    if (e != hipSuccess) return fail(std::string("hipHostRegister of the shared segment failed: ") + hipGetErrorString(e), DCORA_ERR_HIP);
    registered_ = true;
    DCORA_HIP(hipHostGetDevicePointer((void **)&dev_map_, map_, 0));
  }
  DCORA_HIP(arrive_.alloc(R));
  DCORA_HIP(hipMemset(arrive_.p, 0, sizeof(unsigned) * R));
  DCORA_HIP(arrive2_.alloc(R));
  DCORA_HIP(hipMemset(arrive2_.p, 0, sizeof(unsigned) * R));
  {
    char bus[64] = {0};
    uint64_t h = 1469598103934665603ull;
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), s->x_device()) != hipSuccess) {
      (void)hipGetLastError();
      std::snprintf(bus, sizeof(bus), "device-%d", s->x_device());
    }
    for (const char *c = bus; *c; ++c) h = (h ^ (uint64_t)(unsigned char)*c) * 1099511628211ull;
    ranks_[rank].bus.store(h ? h : 1, std::memory_order_release);
  }
  DCORA_HIP(evalbuf_.alloc(2 * R));
  DCORA_HIP(hipMemset(evalbuf_.p, 0, sizeof(double) * 2 * R));
  DCORA_HIP(hosted_list_.alloc(std::max(1, n_hosted_)));
  if (n_hosted_)
    DCORA_HIP(hipMemcpy(hosted_list_.p, hosted.data(), sizeof(int) * n_hosted_, hipMemcpyHostToDevice));

  const char *force = env::exchange();
  const bool want_ipc = !(force && std::strcmp(force, "staged") == 0);
  int ok = 1;
  if (world > 1) {
    rc = setup_ipc(want_ipc);
    if (rc != DCORA_OK && rc != DCORA_ERR_UNSUPPORTED) return rc;
    ok = rc == DCORA_OK;
  }
  ranks_[rank].ipc_ok.store(ok ? 1 : -1, std::memory_order_release);
  int rc2 = barrier();
  if (rc2) return rc2;
  bool all = true;
  for (int q = 0; q < world; ++q) all = all && ranks_[q].ipc_ok.load(std::memory_order_acquire) == 1;
  mode = all ? kExchangeIpc : kExchangeStaged;
  {
    // Who waits for a producer's flag.  With a GPU of its own the consumer's scatter kernel polls the flag itself: no
    // host hop between the producer's stores and the scatter.  Ranks SHARING a GPU (the rehearsal on one device) keep
    // the host's wait: a kernel that polls occupies the queue the producer's kernels of the other process need --
    // measured with 4 ranks on one MI355X: 875 against 1084 it/s on the headline, 274 against 664 on the 100k lattice.
    // DCORA_EXCHANGE_WAIT=device / host forces one or the other.
    bool shared_gpu = false;
    for (int q = 0; q < world; ++q)
      if (q != rank && ranks_[q].bus.load(std::memory_order_acquire) == ranks_[rank].bus.load()) shared_gpu = true;
    // A peer GPU's stores into a COARSE-grained halo buffer are not guaranteed visible to a kernel of this GPU that is
    // already running (lines may be served from the local L2; the host-wait form relies on the invalidate of the
    // kernel boundary): the kernel polls only where every rank's halo is fine-grained device memory.
    bool all_fine = true;
    for (int q = 0; q < world; ++q) all_fine = all_fine && ranks_[q].fine.load(std::memory_order_acquire) == 1;
    const bool poll_is_safe = mode != kExchangeIpc || all_fine;
    const char *wm = env::exchange_wait();
    device_wait_ = (wm ? std::strcmp(wm, "host") != 0 : !shared_gpu) && poll_is_safe;
  }
  if (force && std::strcmp(force, "ipc") == 0 && !all) return fail("DCORA_EXCHANGE=ipc but the IPC transport is not usable", DCORA_ERR_HIP);
  rc = barrier();
  if (rc) return rc;
  if (world > 1) {
    rc = link_check();
    if (rc) return rc;
  }
  rc = barrier();
  if (rc) return rc;
  if (rank == 0) shm_unlink(name_.c_str());  // every rank has it mapped: nothing is left in /dev/shm after the job
  return DCORA_OK;
}

// Halo buffers mapped into every rank that writes to them, then one round of test stores checked by the owner.
// Every rank passes the same two barriers whatever fails locally; the result is this rank's vote.
int Exchange::setup_ipc(bool attempt) {
  const int R = R_;
  const size_t test_off = 2 * (size_t)R * slot_;
  const int ntest = 64;
  devflag_off_ = align_up(test_off + (size_t)ntest * world, 8);
  probe_off_ = devflag_off_ + 2 * (size_t)R * 8;  // behind one 64-byte flag word per (parity, agent)
  const size_t halo_doubles = probe_off_ + (size_t)world * (kProbeDoubles + 8);  // + the link check's area per writer
  bool ok = attempt;
  std::string why;
  auto no = [&](const std::string &m) {
    if (ok) why = m;
    ok = false;
  };
  // The halo buffer is written by OTHER GPUs while this GPU's kernels read it launch after launch: fine-grained device
  // memory (not cached incoherently in this GPU's L2), so that a slot re-used two posts later is never served stale;
  // plain hipMalloc if the runtime refuses (the self-test below still has to pass).
  if (ok) {
    const size_t bytes = sizeof(double) * halo_doubles;
    void *hp = nullptr;
    if (hipExtMallocWithFlags(&hp, bytes, hipDeviceMallocFinegrained) == hipSuccess && hp) {
      halo_.p = (double *)hp;  // released with hipFree like any DevBuf
      halo_.n = halo_doubles;
      halo_finegrained_ = true;
    } else {
      (void)hipGetLastError();
      if (halo_.alloc(halo_doubles) != hipSuccess) no("hipMalloc of the halo buffer");
    }
  }
  if (ok && hipMemset(halo_.p, 0, sizeof(double) * halo_doubles) != hipSuccess) no("hipMemset");
  if (ok && hipIpcGetMemHandle(&ranks_[rank].halo, halo_.p) != hipSuccess) {
    (void)hipGetLastError();
    if (halo_finegrained_) {  // a runtime that exports no handle for fine-grained memory: once more with plain hipMalloc
      halo_.release();
      halo_finegrained_ = false;
      if (halo_.alloc(halo_doubles) != hipSuccess ||
          hipMemset(halo_.p, 0, sizeof(double) * halo_doubles) != hipSuccess ||
          hipIpcGetMemHandle(&ranks_[rank].halo, halo_.p) != hipSuccess)
        no("hipIpcGetMemHandle");
    } else {
      no("hipIpcGetMemHandle");
    }
  }
  (void)hipGetLastError();
  ranks_[rank].device.store(s_->x_device());
  ranks_[rank].pid.store((int)getpid());
  ranks_[rank].fine.store(ok && halo_finegrained_ ? 1 : 0);
  ranks_[rank].published.store(ok ? 1 : -1, std::memory_order_release);
  int rc = barrier();
  if (rc) return rc;
  // whom do I write to: the owners of my hosted agents' neighbours
  std::set<int> peers;
  for (int a = 0; a < R; ++a) peers.insert(dests_[a].begin(), dests_[a].end());
  for (int q : peers) {
    if (!ok) break;
    if (ranks_[q].published.load(std::memory_order_acquire) != 1) {
      no("rank " + std::to_string(q) + " has no halo handle");
      break;
    }
    const int pd = ranks_[q].device.load();
    if (pd != s_->x_device()) {
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, s_->x_device(), pd) == hipSuccess && can) {
        const hipError_t e = hipDeviceEnablePeerAccess(pd, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) no("hipDeviceEnablePeerAccess");
      }
      (void)hipGetLastError();
    }
    void *p = nullptr;
    hipIpcMemHandle_t h;
    std::memcpy(&h, &ranks_[q].halo, sizeof(h));
    if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
      (void)hipGetLastError();
      no("hipIpcOpenMemHandle of rank " + std::to_string(q));
      break;
    }
    peer_halo_[q] = (double *)p;
    opened_[q] = true;
  }
  // self-test: a pattern into my strip of every peer's test area, checked by the owner after the barrier
  if (ok) {
    for (int q : peers)
      hipLaunchKernelGGL(k_selftest_write, dim3(1), dim3(64), 0, s_->x_stream(),
                         peer_halo_[q] + test_off + (size_t)ntest * rank, ntest, 1000.0 * (rank + 1));
    if (hipStreamSynchronize(s_->x_stream()) != hipSuccess) no("self-test stores");
    (void)hipGetLastError();
  }
  rc = barrier();
  if (rc) return rc;
  if (ok) {
    std::vector<double> h((size_t)ntest * world);
    if (hipMemcpy(h.data(), halo_.p + test_off, sizeof(double) * h.size(), hipMemcpyDeviceToHost) != hipSuccess)
      no("reading the self-test area");
    std::set<int> writers;  // ranks hosting an agent whose public poses I need
    for (int a = 0; a < R; ++a)
      if (needed_[a]) writers.insert(owner_[a]);
    for (int q : writers)
      for (int i = 0; i < ntest && ok; ++i)
        if (h[(size_t)ntest * q + i] != 1000.0 * (q + 1) + i) no("self-test pattern of rank " + std::to_string(q) + " did not arrive");
  }
  if (!ok && attempt)
    set_last_error("exchange: IPC transport unavailable (" + why + "), the ranks use the shared host segment");
  return ok ? DCORA_OK : DCORA_ERR_UNSUPPORTED;
}

// One round of the link check: a 4 KB pattern and a flag from every rank into each rank it will write to, through the
// transport (mode) and the form of the wait (device_wait_) in force; true when everything THIS rank reads arrived intact
// within the budget.  Never blocks longer than a few seconds whatever a peer does.
bool Exchange::probe_round(uint64_t seq, std::string *why) {
  const int R = R_;
  const bool ipc = mode == kExchangeIpc;
  std::set<int> peers, writers;
  for (int a = 0; a < R; ++a) {
    peers.insert(dests_[a].begin(), dests_[a].end());
    if (needed_[a]) writers.insert(owner_[a]);
  }
  auto pattern = [&](int writer, int reader) { return 1e6 * (double)seq + 4096.0 * writer + 64.0 * reader; };
  auto hflag_dev = [&](int reader, int writer) {
    return (volatile uint64_t *)(dev_map_ + off_probe_flags_ + sizeof(ShmFlag) * ((size_t)reader * world + writer));
  };
  bool ok = true;
  auto no = [&](const std::string &m) {
    if (ok && why) *why = m;
    ok = false;
  };
  if (hipSetDevice(s_->x_device()) != hipSuccess) no("hipSetDevice");
  hipStream_t st = s_->x_stream();
  for (int q : peers) {
    if (!ok) break;
    double *dst = ipc ? peer_halo_[q] + probe_off_ + (size_t)rank * (kProbeDoubles + 8)
                      : (double *)(dev_map_ + off_probe_stage_) + ((size_t)q * world + rank) * kProbeDoubles;
    uint64_t *dflag = ipc ? (uint64_t *)(peer_halo_[q] + probe_off_ + (size_t)rank * (kProbeDoubles + 8) + kProbeDoubles) : nullptr;
    if (ipc && !peer_halo_[q]) {
      no("no mapping of rank " + std::to_string(q) + "'s halo buffer");
      break;
    }
    hipLaunchKernelGGL(k_probe_post, dim3(1), dim3(256), 0, st, dst, dflag, hflag_dev(q, rank), pattern(rank, q), seq);
  }
  if (hipGetLastError() != hipSuccess) no("probe post launch");
  const double budget_s = 3.0;
  for (int p : writers) {
    if (!ok) break;
    const ShmFlag *hf = (const ShmFlag *)((char *)map_ + off_probe_flags_) + ((size_t)rank * world + p);
    if (!device_wait_) {  // the host waits for the flag, as wait_arr does
      const auto w0 = Clock::now();
      unsigned spins = 0;
      while (hf->seq < seq) {
        polite_spin(spins);
        if ((spins & 255u) == 0 && (hdr_->failed.load() || since(w0) > budget_s)) {
          no("the probe flag of rank " + std::to_string(p) + " never arrived (host wait)");
          break;
        }
      }
      std::atomic_thread_fence(std::memory_order_acquire);
      if (!ok) break;
    }
    const double *src = ipc ? halo_.p + probe_off_ + (size_t)p * (kProbeDoubles + 8)
                            : (const double *)(dev_map_ + off_probe_stage_) + ((size_t)rank * world + p) * kProbeDoubles;
    const uint64_t *flag = ipc ? (const uint64_t *)(halo_.p + probe_off_ + (size_t)p * (kProbeDoubles + 8) + kProbeDoubles)
                               : (const uint64_t *)hflag_dev(rank, p);
    int *res = (int *)(dev_map_ + off_probe_res_ + sizeof(ShmFlag) * ((size_t)rank * world + p));
    hipLaunchKernelGGL(k_probe_wait, dim3(1), dim3(256), 0, st, src, flag, seq, (long long)(budget_s * 1e8),
                       device_wait_ ? 1 : 0, pattern(p, rank), res);
  }
  if (hipGetLastError() != hipSuccess) no("probe wait launch");
  // the stream drains within the kernels' own budgets; a bounded host wait on top (never hipStreamSynchronize blindly)
  {
    const auto w0 = Clock::now();
    hipError_t e;
    while ((e = hipStreamQuery(st)) == hipErrorNotReady) {
      usleep(200);
      if (since(w0) > 4.0 * budget_s) {
        no("the probe kernels did not finish");
        break;
      }
    }
    if (e != hipSuccess && e != hipErrorNotReady) {
      (void)hipGetLastError();
      no(std::string("probe kernels: ") + hipGetErrorString(e));
    }
  }
  for (int p : writers) {
    if (!ok) break;
    const volatile int *res = (const volatile int *)((char *)map_ + off_probe_res_ + sizeof(ShmFlag) * ((size_t)rank * world + p));
    if (*res != 1)
      no("probe of rank " + std::to_string(p) + (*res == 2 ? ": the flag never arrived" : *res == 3 ? ": the 4 KB pattern arrived damaged" : ": no result"));
    *(volatile int *)res = 0;
  }
  return ok;
}

// Start-up check of the links this rank will use (dcora_hip.h, dcora_exchange_info): rounds of probe_round, all ranks
// stepping down together -- device-side wait -> host wait -> staged transport -- until a round passes on every rank.
int Exchange::link_check() {
  std::string why, first_why;
  // test hook (dcora_debug_exchange_probe_fault): the last rank reports its first n rounds as failed
  const int fault_rounds = g_probe_fault_rounds.load();
  for (int round = 1; round <= 3; ++round) {
    const auto t0 = Clock::now();
    bool mine = probe_round((uint64_t)round, &why);
    if (mine && rank == world - 1 && round <= fault_rounds) {
      mine = false;
      why = "injected fault (dcora_debug_exchange_probe_fault)";
    }
    if (!mine && first_why.empty()) first_why = why;
    // the vote carries the form of the wait this rank used (device_wait_ is rank-local: only ranks with a GPU of their
    // own poll on the device), so that the step down below is taken from SHARED state and is the same on every rank
    const int vote = 2 * round + (device_wait_ ? 1 : 0);
    ranks_[rank].probe.store(mine ? vote : -vote, std::memory_order_release);
    int rc = barrier(30.0);
    if (rc) return rc;
    bool all = true, any_device_wait = false;
    for (int q = 0; q < world; ++q) {
      const int v = ranks_[q].probe.load(std::memory_order_acquire);
      all = all && v > 0 && v / 2 == round;
      any_device_wait = any_device_wait || ((v < 0 ? -v : v) & 1);
    }
    rc = barrier(30.0);  // nobody overwrites its vote before everybody has read the votes
    if (rc) return rc;
    link_rounds = round;
    link_last_us = 1e6 * since(t0);
    if (all) {
      if (round > 1)
        set_last_error("exchange: link check passed after stepping down (" + first_why + "): " +
                       (mode == kExchangeIpc ? "IPC peer stores, host wait" : "shared host segment"));
      return DCORA_OK;
    }
    // the same step on every rank: it depends on the votes and on `mode` only (identical everywhere), never on this
    // rank's own device_wait_ -- ranks that share a GPU wait on the host from the start while the others poll on the
    // device, and a ladder keyed on the local flag sent them to different rungs (mixed transports, a spurious
    // DCORA_ERR_EXCHANGE_LINK): device wait anywhere -> host wait everywhere -> staged with host wait everywhere
    if (any_device_wait) {
      if (device_wait_) link_gave_up_device_wait = 1;
      device_wait_ = false;
    } else if (mode == kExchangeIpc) {
      const char *force = env::exchange();
      if (force && std::strcmp(force, "ipc") == 0) break;  // (the environment of a job is the same on its ranks)
      mode = kExchangeStaged;
      link_gave_up_ipc = 1;
    } else {
      break;
    }
  }
  return fail("link check: no transport between the ranks works (" + (first_why.empty() ? why : first_why) + ")",
              DCORA_ERR_EXCHANGE_LINK);
}

int Exchange::post(const int *agents, int count) { return post_arr(agents, count, s_->x_rank_r(), s_->x_mirror()); }
int Exchange::wait(const int *agents, int count) { return wait_arr(agents, count, s_->x_rank_r(), s_->x_mirror()); }

// the same exchange for any r x (d+1)n array laid out like X (the certificate's vectors: r = 1)
int Exchange::post_arr(const int *agents, int count, int r, const double *arr) {
  const auto t0 = Clock::now();
  DCORA_HIP(hipSetDevice(s_->x_device()));
  const int R = R_;
  for (int i = 0; i < count; ++i) {
    const int a = agents[i];
    if (a < 0 || a >= R) return usage("post: agent out of range", DCORA_ERR_BAD_ARG);
    const uint64_t q = ++seq_[a];
    const XAgentView ag = s_->x_agent(a);
    if (!ag.hosted || dests_[a].empty() || ag.ncols == 0) continue;
    const int parity = (int)(q & 1);
    // Back-pressure: the slot of this parity was last written by post q - 2; every rank that reads it must have
    // scattered that post before it is overwritten (the evaluation's heartbeat used to be the only thing between a
    // producer running ahead and a torn read: ticks of one set posted again and again, the per-phase C ABI).
    if (q > 2) {
      const auto b0 = Clock::now();
      for (int p : dests_[a]) {
        const ShmFlag *cf = consumed_ + (size_t)p * R + a;
        unsigned spins = 0;
        while (cf->seq + 2 < q) {
          polite_spin(spins);
          if ((spins & 1023u) == 0) {
            if (hdr_->failed.load()) return fail("another rank failed", DCORA_ERR_HIP);
            if (since(b0) > exchange_timeout_s())
              return fail("rank " + std::to_string(p) + " never read post " + std::to_string(q - 2) + " of agent " +
                              std::to_string(a),
                          DCORA_ERR_HIP);
          }
        }
      }
    }
    PostDst dst{}, dflag{};
    if (mode == kExchangeIpc) {
      for (int p : dests_[a]) {
        dst.base[dst.n++] = peer_halo_[p] + halo_off(parity, a);
        dflag.base[dflag.n++] = peer_halo_[p] + devflag_off_ + ((size_t)parity * R + a) * 8;
      }
    } else {
      dst.base[dst.n++] = (double *)(dev_map_ + off_staged_) + halo_off(parity, a);
    }
    const int ncols = ag.ncols;
    const long N = (long)ncols * r;
    const int grid = (int)std::min<long>((N + kBlock - 1) / kBlock, 64);
    volatile uint64_t *flag = (volatile uint64_t *)(dev_map_ + off_flags_ + sizeof(ShmFlag) * ((size_t)parity * R + a));
    hipLaunchKernelGGL(k_post_public, dim3(grid), dim3(kBlock), 0, s_->x_stream(), r, ncols, ag.cols_dev, arr, dst,
                       dflag, arrive_.p + a, flag, q);
    bytes_posted += 8.0 * N * dst.n;
    ++posts;
  }
  DCORA_HIP(hipGetLastError());
  post_s += since(t0);
  return DCORA_OK;
}

int Exchange::wait_arr(const int *agents, int count, int r, double *arr) {
  const auto t0 = Clock::now();
  DCORA_HIP(hipSetDevice(s_->x_device()));
  const int R = R_;
  for (int i = 0; i < count; ++i) {
    const int a = agents[i];
    if (a < 0 || a >= R) return usage("wait: agent out of range", DCORA_ERR_BAD_ARG);
    const XAgentView ag = s_->x_agent(a);
    if (!needed_[a] || ag.ncols == 0) continue;
    const uint64_t want = seq_[a];
    const int parity = (int)(want & 1);
    const ShmFlag *f = flags_ + (size_t)parity * R + a;
    if (!device_wait_) {  // DCORA_EXCHANGE_WAIT=host: the host waits for the flag, the kernel below finds it set
      unsigned spins = 0;
      const auto w0 = Clock::now();
      while (f->seq < want) {
        polite_spin(spins);
        if ((spins & 1023u) == 0) {
          if (hdr_->failed.load()) return fail("another rank failed", DCORA_ERR_HIP);
          if (since(w0) > exchange_timeout_s())
            return fail("public poses of agent " + std::to_string(a) + " never arrived", DCORA_ERR_HIP);
        }
      }
      std::atomic_thread_fence(std::memory_order_acquire);
    }
    const bool ipc = mode == kExchangeIpc;
    const double *src = ipc ? halo_.p + halo_off(parity, a) : (const double *)(dev_map_ + off_staged_) + halo_off(parity, a);
    // the flag the kernel polls: next to the slot in this rank's halo buffer (stored by the producer over xGMI), or the
    // host-visible word when the poses are staged in the shared segment
    const uint64_t *dflag = ipc ? (const uint64_t *)(halo_.p + devflag_off_ + ((size_t)parity * R + a) * 8)
                                : (const uint64_t *)(dev_map_ + off_flags_ + sizeof(ShmFlag) * ((size_t)parity * R + a));
    const int ncols = ag.ncols;
    const long N = (long)ncols * r;
    const int grid = (int)std::min<long>((N + kBlock - 1) / kBlock, 32);
    uint64_t *cons = (uint64_t *)(dev_map_ + off_consumed_ + sizeof(ShmFlag) * ((size_t)rank * R + a));
    uint32_t *failed = (uint32_t *)(dev_map_ + offsetof(ShmHeader, failed));
    // updateNeighborStates: into the local mirror of X
    hipLaunchKernelGGL(k_wait_scatter, dim3(grid), dim3(kBlock), 0, s_->x_stream(), r, ncols, ag.cols_dev, src,
                       arr, dflag, want, arrive2_.p + a, cons, failed);
    ++waits;
  }
  wait_s += since(t0);
  return DCORA_OK;
}

// distributed form of the central evaluation (ref examples/MultiRobotExample.cpp:264-305): every rank evaluates the
// agents it hosts against the neighbours' public poses it holds, publishes two scalars per agent, reads everybody's
int Exchange::evaluate(double *cost2, double *gradnorm, double *block_norms, int *next_selected) {
  const int R = R_;
  int rc = s_->x_phase_evaluate_dev(evalbuf_.p);
  if (rc) return rc;
  const uint64_t want = ++eval_seq_;
  const int parity = (int)(want & 1);
  const size_t per_parity = (size_t)R + world;
  ShmEval *slots_dev = (ShmEval *)(dev_map_ + off_evals_) + (size_t)parity * per_parity;
  if (n_hosted_)
    hipLaunchKernelGGL(k_eval_publish, dim3(1), dim3(64), 0, s_->x_stream(), n_hosted_, hosted_list_.p, evalbuf_.p, slots_dev,
                       want);
  DCORA_HIP(hipGetLastError());
  const auto t0 = Clock::now();
  ShmEval *sl = evals_ + (size_t)parity * per_parity;
  // Heartbeat of this rank: every rank -- also one that hosts no agent and therefore publishes nothing -- says that
  // it has entered evaluation `want`, and nobody leaves it before all have.  A rank can then never be lapped: the
  // slot of parity `want` is overwritten at evaluation want + 2, which every writer enters only after all ranks have
  // entered want + 1, i.e. after they have finished reading `want`.
  std::atomic_thread_fence(std::memory_order_release);
  sl[R + rank].seq = want;
  for (int q = 0; q < world; ++q) {
    unsigned spins = 0;
    while (sl[R + q].seq < want) {
      polite_spin(spins);
      if ((spins & 1023u) == 0) {
        if (hdr_->failed.load()) return fail("another rank failed", DCORA_ERR_HIP);
        if (since(t0) > exchange_timeout_s()) return fail("rank " + std::to_string(q) + " never entered the evaluation", DCORA_ERR_HIP);
      }
    }
  }
  double g2 = 0, c2 = 0, best = -1;
  int arg = 0;
  for (int a = 0; a < R; ++a) {
    unsigned spins = 0;
    while (sl[a].seq < want) {
      polite_spin(spins);
      if ((spins & 1023u) == 0) {
        if (hdr_->failed.load()) return fail("another rank failed", DCORA_ERR_HIP);
        if (since(t0) > exchange_timeout_s()) return fail("evaluation of agent " + std::to_string(a) + " never arrived", DCORA_ERR_HIP);
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    const double ga = sl[a].g2, xa = sl[a].xeg;
    const double nb = std::sqrt(ga);
    if (block_norms) block_norms[a] = nb;
    g2 += ga;
    c2 += xa;  // 2 f = sum_b <X_b, X_b Q_bb + G_b>
    if (nb > best) {
      best = nb;
      arg = a;
    }
  }
  eval_wait_s += since(t0);
  ++evals;
  if (cost2) *cost2 = c2;
  if (gradnorm) *gradnorm = std::sqrt(g2);
  if (next_selected) *next_selected = arg;
  return DCORA_OK;
}

// one pass of the reference driver's loop body (examples/MultiRobotExample.cpp:223-307) across the ranks
int Exchange::rbcd_iterate(int selected, double *cost2, double *gradnorm, double *block_norms, int *next_selected) {
  const int R = R_;
  if (selected < 0 || selected >= R) return usage("selected agent out of range", DCORA_ERR_BAD_ARG);
  std::vector<int> others;
  for (int a = 0; a < R; ++a)
    if (a != selected) others.push_back(a);
  int rc = s_->x_phase_nonselected(selected);  // Agent::iterate(false) of the hosted non-selected agents
  if (rc) return rc;
  rc = post(others.data(), (int)others.size());
  if (rc) return rc;
  rc = wait(others.data(), (int)others.size());  // the selected agent's pull (and everybody's for the evaluation)
  if (rc) return rc;
  rc = s_->x_phase_selected(selected);  // Agent::iterate(true) where the selected agent lives
  if (rc) return rc;
  rc = post(&selected, 1);
  if (rc) return rc;
  rc = wait(&selected, 1);
  if (rc) return rc;
  int nxt = selected;
  rc = evaluate(cost2, gradnorm, block_norms, &nxt);
  if (rc) return rc;
  if (next_selected) *next_selected = s_->x_agent(selected).neighbors->empty() ? selected : nxt;
  return DCORA_OK;
}

int Exchange::rbcd_tick(const int *set, int count, int allow_adjacent) {
  int rc = s_->x_iterate_set(set, count, allow_adjacent);
  if (rc) return rc;
  rc = post(set, count);
  if (rc) return rc;
  return wait(set, count);
}

int Exchange::set_X(const double *Xh) {
  int rc = barrier();
  if (rc) return rc;
  rc = s_->x_set_X(Xh);
  if (rc) return rc;
  return barrier();
}

// every rank's hosted blocks -> the shared segment -> every rank's copy of the whole X
int Exchange::gather_X(double *Xh) {
  int rc = s_->x_stage_hosted(xarea_);
  if (rc) return rc;
  rc = barrier();
  if (rc) return rc;
  std::memcpy(Xh, xarea_, sizeof(double) * (size_t)s_->x_rank_r() * (size_t)s_->x_num_cols());
  return barrier();
}

int Exchange::debug_leave_stale(const char *job_name, int world_, int R) {
  rank = 0;
  world = world_;
  R_ = R;
  slot_ = 16;
  const int rc = map_segment(job_name, 16);
  if (rc) return rc;
  hdr_->creator_pid = 0x7ffffff0;  // beyond any pid_max: nobody answers
  hdr_->bar_count.store(1);        // and its barrier stands as the crash left it
  name_.clear();                   // the destructor must not remove it
  return DCORA_OK;
}

// The host half of the protocol on its own (no device): bootstrap through the segment, barriers, and `rounds` rounds
// of post -> wait -> evaluation all-gather in which host stores stand in for the device's (same slots, same parity
// double-buffering, same sequence numbers).  checksum is identical on every rank.
int Exchange::host_selftest(const char *job_name, int rank_, int world_, int R, int rounds, double *checksum) {
  rank = rank_;
  world = world_;
  R_ = R;
  if (world < 1 || rank < 0 || rank >= world || world > kMaxRanks || R < 1 || R > kMaxAgents || rounds < 1)
    return fail("host selftest: bad arguments", DCORA_ERR_BAD_ARG);
  slot_ = 16;
  int rc = map_segment(job_name, 16);
  if (rc) return rc;
  rc = barrier();
  if (rc) return rc;
  if (rank == 0) shm_unlink(name_.c_str());
  const int per = (R + world - 1) / world;
  double sum = 0;
  for (int q = 1; q <= rounds; ++q) {
    const int parity = q & 1;
    for (int a = 0; a < R; ++a) {
      if (a / per != rank) continue;
      double *dst = staged_ + ((size_t)parity * R + a) * slot_;
      for (size_t i = 0; i < slot_; ++i) dst[i] = 1000.0 * q + 16.0 * a + (double)i;
      std::atomic_thread_fence(std::memory_order_release);
      flags_[(size_t)parity * R + a].seq = (uint64_t)q;
    }
    const auto t0 = Clock::now();
    for (int a = 0; a < R; ++a) {
      const ShmFlag *f = flags_ + (size_t)parity * R + a;
      unsigned spins = 0;
      while (f->seq < (uint64_t)q) {
        if ((++spins & 1023u) == 0) {
          if (hdr_->failed.load()) return fail("another rank failed", DCORA_ERR_HIP);
          if (since(t0) > exchange_timeout_s()) return fail("host selftest: a post never arrived", DCORA_ERR_HIP);
          sched_yield();
        }
      }
      std::atomic_thread_fence(std::memory_order_acquire);
      const double *src = staged_ + ((size_t)parity * R + a) * slot_;
      for (size_t i = 0; i < slot_; ++i)
        if (src[i] != 1000.0 * q + 16.0 * a + (double)i) return fail("host selftest: payload mismatch", DCORA_ERR_HIP);
    }
    const size_t per_parity = (size_t)R + world;
    for (int a = 0; a < R; ++a) {
      if (a / per != rank) continue;
      ShmEval *e = evals_ + (size_t)parity * per_parity + a;
      e->g2 = q + 0.5 * a;
      e->xeg = q * 0.25 - a;
      std::atomic_thread_fence(std::memory_order_release);
      e->seq = (uint64_t)q;
    }
    {  // the rank's heartbeat (see Exchange::evaluate): a rank without agents must not be lapped either
      ShmEval *hb = evals_ + (size_t)parity * per_parity + R;
      std::atomic_thread_fence(std::memory_order_release);
      hb[rank].seq = (uint64_t)q;
      for (int p2 = 0; p2 < world; ++p2) {
        unsigned spins = 0;
        while (hb[p2].seq < (uint64_t)q) {
          if ((++spins & 1023u) == 0) {
            if (hdr_->failed.load()) return fail("another rank failed", DCORA_ERR_HIP);
            if (since(t0) > exchange_timeout_s()) return fail("host selftest: a heartbeat never arrived", DCORA_ERR_HIP);
            sched_yield();
          }
        }
      }
    }
    for (int a = 0; a < R; ++a) {
      const ShmEval *e = evals_ + (size_t)parity * per_parity + a;
      unsigned spins = 0;
      while (e->seq < (uint64_t)q) {
        if ((++spins & 1023u) == 0) {
          if (hdr_->failed.load()) return fail("another rank failed", DCORA_ERR_HIP);
          if (since(t0) > exchange_timeout_s()) return fail("host selftest: an evaluation never arrived", DCORA_ERR_HIP);
          sched_yield();
        }
      }
      std::atomic_thread_fence(std::memory_order_acquire);
      sum += e->g2 * (a + 1) + e->xeg;
    }
  }
  if (checksum) *checksum = sum;
  return barrier();
}

}  // namespace dcora
