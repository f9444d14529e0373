// Neighbour exchange of public poses between the ranks of ONE node (one process per GPU), inside the library
// (replaces the transport the reference leaves to its host: Agent::getSharedStateDicts -> updateNeighborStates,
// ref src/Agent.cpp:113-152, 844-906, driven by examples/MultiRobotExample.cpp:236-258).
//
// Transport: every rank owns a halo buffer in its HBM with one slot per agent; the peers map
// it through HIP IPC.  A post is ONE kernel per hosted agent: it gathers the agent's public poses from the mirror of
// X and stores them straight into the halo slot of every rank that hosts a neighbour of that agent (peer stores over
// xGMI), fences at system scope and lets its last workgroup store the agent's sequence number into a flag word.  Flag
// words and the 2R evaluation scalars live in a POSIX shared-memory segment that every rank registers with HIP
// (device-writable, host-readable): a device store there IS the all-gather.  The receiving host spins on the flag
// word and enqueues the scatter into its mirror.  No collective, no ring, no host copy of pose data.
// When IPC mapping (or its self-test) fails on any rank, all ranks fall back to staging the packed poses in the same
// shared host segment (device store over PCIe, device load over PCIe on the consumer).
#pragma once
#include <atomic>
#include <cstdint>
#include <string>
#include <vector>

#include "rbcd.h"

namespace dcora {

constexpr int kMaxRanks = 64;
constexpr uint32_t kShmMagic = 0x44434f52u;  // "DCOR"
constexpr int kProbeDoubles = 512;           // the link check's payload: 4 KB

struct alignas(64) ShmFlag {
  volatile uint64_t seq;
  uint64_t pad[7];
};
struct alignas(64) ShmEval {
  volatile double g2, xeg;  // |Proj(X_b Q_bb + G_b)|^2, <X_b, X_b Q_bb + G_b>
  volatile uint64_t seq;
  uint64_t pad[5];
};
struct alignas(64) ShmRank {
  hipIpcMemHandle_t halo;  // 64 bytes
  std::atomic<int> device, pid, ipc_ok, published;
  std::atomic<uint64_t> bus;  // hash of the device's PCI bus id: two ranks with the same value share a GPU
  std::atomic<int> fine;      // 1: this rank's halo buffer is fine-grained device memory
  std::atomic<int> probe;     // link check: +round passed, -round failed
  uint64_t pad[4];
};
struct alignas(64) ShmRed {  // one rank's contribution to a sum over the ranks
  volatile uint64_t seq;
  double vals[31];
};
struct ShmHeader {
  std::atomic<uint32_t> magic;
  uint32_t world, R;
  uint64_t slot_doubles, total_bytes, x_doubles;
  std::atomic<uint32_t> bar_count, bar_gen;
  std::atomic<uint32_t> failed;  // a rank gave up: everybody waiting returns an error instead of spinning on
  int32_t creator_pid;           // rank 0's process: a segment whose creator is gone is a stale one (crashed job)
  uint64_t creator_start;        // ... and its start time (/proc/<pid>/stat): a recycled pid is not the creator
};

enum ExchangeMode { kExchangeIpc = 1, kExchangeStaged = 2 };

// how long a rank waits for another before it gives up, raises `failed` for everybody and returns an error
// (DCORA_EXCHANGE_TIMEOUT_S, default 120: a rank that died takes the job down within this time, never a hang)
double exchange_timeout_s();
extern std::atomic<int> g_probe_fault_rounds;  // test hook of the link check (dcora_debug_exchange_probe_fault)

class Exchange {
 public:
  ~Exchange();
  int init(ExchangeSession *s, const char *job_name);
  int post(const int *agents, int count);
  int wait(const int *agents, int count);
  int post_arr(const int *agents, int count, int r, const double *arr);
  int wait_arr(const int *agents, int count, int r, double *arr);
  // sum of `count` (<= 31) doubles over the ranks, added in rank order: the same bits on every rank
  int allreduce_sum(double *vals, int count);
  // Certification across the ranks ("Collective"): fastVerification (ref src/DCORA_utils.cpp:1713-1735)
  // of the current iterate.  The PSD test runs on rank 0 (it assembles S from the gathered X and the global Q it is
  // given; Qglobal may be null on the other ranks); when it fails, the minimum eigenpair of S + eta I is computed by
  // ALL ranks: each applies its row block of S (its agents' Q_bb and coupling blocks, the Lambda blocks of its poses)
  // to its slice of the Lanczos vectors, the public entries travel like public poses, every inner product is an
  // allreduce_sum.  v (may be null): the eigenvector, (d+1) n doubles, the same on every rank.
  int certify(const HostCsr *Qglobal, double eta, int *certified, double *theta, double *lambda_min, double *v,
              long long *matvecs, int *distributed);
  int evaluate(double *cost2, double *gradnorm, double *block_norms, int *next_selected);
  int rbcd_iterate(int selected, double *cost2, double *gradnorm, double *block_norms, int *next_selected);
  int rbcd_tick(const int *set, int count, int allow_adjacent);
  int barrier(double timeout_s = 120.0);
  int gather_X(double *Xh);
  // Agent::setX of every agent on every rank: sequence numbers restart with the Nesterov sequences
  int set_X(const double *Xh);
  // host half of the protocol alone (bootstrap, barriers, flags, evaluation slots; host stores stand in for the
  // device's): runs without a GPU, for the world-size-2 CPU test
  int host_selftest(const char *job_name, int rank, int world, int R, int rounds, double *checksum);
  // test hook: what a crashed job of this shape leaves behind under the name (initialised, creator gone)
  int debug_leave_stale(const char *job_name, int world, int R);

  int mode = 0;
  int rank = 0, world = 1;
  // link check (init): rounds run, whether the device-side wait / the IPC transport were given up, time of the last round
  int link_rounds = 0, link_gave_up_device_wait = 0, link_gave_up_ipc = 0;
  double link_last_us = 0;
  bool halo_is_finegrained() const { return halo_finegrained_; }
  // statistics since creation (host wall time spent in post / wait / the evaluation all-gather, bytes posted)
  bool waits_on_device() const { return device_wait_; }
  double post_s = 0, wait_s = 0, eval_wait_s = 0;
  long posts = 0, waits = 0, evals = 0;
  double bytes_posted = 0;
  int num_peers() const;

 private:
  ExchangeSession *s_ = nullptr;
  std::string name_;
  void *map_ = nullptr;
  size_t map_bytes_ = 0;
  bool registered_ = false;
  char *dev_map_ = nullptr;  // device view of the segment
  ShmHeader *hdr_ = nullptr;
  ShmRank *ranks_ = nullptr;
  ShmFlag *flags_ = nullptr;  // [parity][agent]
  ShmEval *evals_ = nullptr;  // [parity][agent]
  double *staged_ = nullptr;  // [parity][agent][slot]
  double *xarea_ = nullptr;   // r x (d+1) n
  ShmRed *red_ = nullptr;        // [parity][rank]
  size_t off_red_ = 0;
  uint64_t red_seq_ = 0;
  ShmFlag *consumed_ = nullptr;  // [consumer rank][agent]: the last post of the agent that rank has scattered
  size_t off_flags_ = 0, off_evals_ = 0, off_staged_ = 0, off_x_ = 0, off_consumed_ = 0;
  size_t off_probe_flags_ = 0, off_probe_res_ = 0, off_probe_stage_ = 0;  // link check: [reader][writer] words / 4 KB
  size_t probe_off_ = 0;  // in a halo buffer: [writer] x (kProbeDoubles payload + 8 doubles of flag)
  size_t devflag_off_ = 0;       // in a halo buffer, behind the slots and the self-test area: [parity][agent] x 64 bytes
  bool device_wait_ = true;      // the scatter kernel polls the flag itself (no host hop between post and scatter)
  DevBuf<unsigned> arrive2_;     // last-workgroup counters of the scatter kernels
  int R_ = 0;
  size_t slot_ = 0;                // doubles per agent slot
  DevBuf<double> halo_;            // [parity][agent][slot] + self-test area (fine-grained device memory)
  bool halo_finegrained_ = false;
  double *peer_halo_[kMaxRanks]{};  // IPC mappings (null for myself and for ranks I never write to)
  bool opened_[kMaxRanks]{};
  DevBuf<unsigned> arrive_;        // one last-workgroup counter per agent
  DevBuf<double> evalbuf_;         // 2R
  DevBuf<int> hosted_list_;
  int n_hosted_ = 0;
  std::vector<int> owner_;               // rank hosting agent a
  std::vector<std::vector<int>> dests_;  // for a hosted agent: the other ranks hosting one of its neighbours
  std::vector<char> needed_;             // agent a (hosted elsewhere) is a neighbour of an agent hosted here
  std::vector<uint64_t> seq_;            // posts of agent a so far (identical on every rank)
  uint64_t eval_seq_ = 0;

  size_t halo_off(int parity, int agent) const { return ((size_t)parity * R_ + agent) * slot_; }
  int open_segment(const char *job_name, size_t bytes);
  int map_segment(const char *job_name, size_t x_doubles);
  int setup_ipc(bool attempt);
  int link_check();
  bool probe_round(uint64_t seq, std::string *why);
  int fail(const std::string &msg, int code);   // a transport / peer failure: raises `failed` for every rank
  int usage(const std::string &msg, int code);  // a refused call (bad argument, unsupported): this call only
};

}  // namespace dcora
