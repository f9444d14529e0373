// Sparse preconditioner application on the device for large blocks:  Z = R (Q + reg I)^-1  without ever forming
// the dense inverse (replaces the CHOLMOD solve of ref src/QuadraticProblem.cpp:70-84 when k is large; small
// blocks keep the dense inverse of device_problem.hip, which is faster while k^2 doubles stay cheap to stream).
//
// Method: nested-dissection Cholesky  P A P^T = L L^T  on the host (once per Q), then L is written as the ordered
// product of its block columns  L = L_1 L_2 ... L_N  (one factor per dissection piece: a leaf sub-domain or a
// separator).  Each factor has an explicit inverse with the same block pattern,
//     L_s^-1 = I + [ D_s^-1 - I ; -B_s D_s^-1 ]  on the columns of piece s,
// and factors of pieces that do not feed each other commute, so all pieces of one level of the dissection tree are
// applied by ONE launch: every output row is an independent gather (dot products of stored weights with the
// current vector), no atomics, fixed summation order => bitwise reproducible.  A solve is
//     permute-in, one launch per group of tree levels (forward), the top, one per group (backward), permute-out
// with depth ~ log2(k / leaf) instead of the thousands of dependent steps of a row-level triangular solve (how levels
// are grouped and the diagonal blocks leave the dependent chain: host_partinv3.cpp).  Vectors ping-pong between two
// buffers per piece; which buffer holds the current value of a piece is known at build time and baked into the gather
// offsets.
#pragma once
#include <cstddef>
#include <memory>
#include <vector>

#include "host_sparse.h"

namespace dcora {

constexpr int kSpTile = 4;  // output rows per tile: they share every gathered vector entry

// ---- matrix-pipe schedule (host_partinv3.cpp, k_sp_mtile): the tiles of the merged schedule -- up to kSpTile consecutive
// output rows that gather from the same sources -- with the sums over the gathered entries inside
// v_mfma_f64_4x4x4_4b_f64: one instruction multiplies FOUR independent 4 x 4 blocks, D_b += A_b B_b; a step of a tile
// feeds it four consecutive groups of four K entries (A_b = the tile's 4 rows x group b of weights, B_b = those 4
// entries x 4 of the r vector values), so a step consumes 16 entries = 512 bytes of weights, ceil(r / 8) * 2 instructions
// of ~17 clocks, and the four partial blocks are added by two DPP row rotations at the end of the tile.
// The matrices a piece owns -- W (or the merged V) and M -- are stored ONCE, in 4 x 4 micro-blocks
//     element (a, e) of an nr x nc matrix  ->  ((a / 4) ncb + e / 4) 16 + (e % 4) 4 + a % 4,      ncb = ceil(nc / 4)
// (rows and columns padded to multiples of four with zeros) -- a micro-block IS one A_b -- and read two ways:
//   kind 0, direct:      A[i][k] = Mat[loc[i]][4 g + k]       K runs over the columns   (forward: rows of W by destination)
//   kind 1, transposed:  A[i][k] = Mat[4 g + k][col0 + i]     K runs over the rows      (backward: W^T; M, symmetric)
// so the backward sweep streams the SAME weights the forward sweep did: stored weights = nnz(W) + nnz(M) where the
// tile schedule of round 3 stored 2 nnz(W) + nnz(M), laid out per tile.
// The unit the device reads is a WAVE RECORD (host_partinv3.cpp packs them, kMtWaves per workgroup): the contiguous range
// of a tile's steps one wave executes -- cut so that it touches at most two segments, whose descriptions travel in the
// record --, where the tile's result goes and which waves of the workgroup hold the other partial sums of the tile.  One
// scalar load tells a wave everything; all loads of up to eight steps are requested together whatever segment they
// belong to.
struct MSub {      // a run of 16-entry steps of one segment
  long long w;     // offset of the matrix in the weight array (multiple of 16)
  int ncb;         // micro-block columns of the matrix
  int ng;          // groups of four K entries of the segment
  int src;         // kind 0: first unknown of the contiguous run the K entries are; kind 1: offset of the index list
  int s0;          // first step of the run
  int n;           // steps of the run
  int pad;
  int loc[4];      // kind 0: the matrix row of every tile row (-1: none); kind 1: loc[0] = col0 (multiple of 4)
};
static_assert(sizeof(MSub) == 48, "MSub is read as three 16-byte words");
struct MWave {
  int out;         // first destination, in unknowns (buffer * k + permuted row); written by the tile's first wave
  int carry;       // first old value to add (same units) or -1
  int nrows;       // <= kSpTile; 0: the wave has nothing to do
  int kind;        // 0: direct weights, contiguous vector run; 1: transposed weights, gathered vector entries
  int red_first;   // wave of the workgroup that holds the tile's first partial sum (adds the others and stores)
  int red_n;       // waves of the tile (consecutive)
  int next;        // record that continues this wave's work (tiles of many short segments), -1: none
  int solo;        // 1: no tile of this workgroup spans several waves (the same in its kMtWaves records): no LDS, no barrier
  MSub a, b;
};
static_assert(sizeof(MWave) == 128, "MWave is read as eight 16-byte words");
constexpr int kMtWaves = 8;  // waves per workgroup of the matrix-pipe kernel

struct SpLevel {  // one launch of the replay
  int task0 = 0;    // first wave record of the launch (PartInvHost::mwaves)
  int ntasks = 0;   // its workgroups (of kMtWaves records each); continuation records follow the launch's workgroups
  int ntiles = 0;   // tiles of the launch (reporting)
  double avg_entries = 0;  // vector entries a tile gathers, on average (reporting)
};

// A host address range whose content also lives on the device (the factor's panels as the device factorisation left
// them, the pieces' M = D^-T D^-1): the builder names its sources by host address, a sink that fills on the device
// resolves them through these ranges.  `host` may be a reserved, unreadable range (the M of device-inverted pieces
// never come to the host).
struct MirrorRange {
  const double *host = nullptr;
  long long n = 0;
  const double *dev = nullptr;
};
namespace partinv {
struct Fill;
}

// Receiver of the stored weights of a schedule, in ascending chunks (write_weights, host_partinv.cpp)
struct WeightSink {
  virtual ~WeightSink() = default;
  // Optional: form ALL weights on the device from the fill records (sources resolved through `mirrors`, the others
  // staged by the sink).  false: not supported / not applicable -- the caller streams chunks as below.
  virtual bool fill_on_device(const std::vector<partinv::Fill> &, long long /*total*/,
                              const std::vector<MirrorRange> & /*mirrors*/, int /*nthreads*/) {
    return false;
  }
  virtual bool wants_device_sources() const { return false; }
  virtual bool begin(long long total) = 0;
  virtual long long chunk_cap() const = 0;                  // most weights one chunk may hold
  virtual double *acquire(long long n) = 0;                 // host memory for the next n <= chunk_cap() weights
  virtual bool commit(long long off, long long n) = 0;      // the memory handed out last holds weights [off, off + n)
  virtual bool end() = 0;
};

// host image of the partitioned inverse: built by build_partitioned_inverse, uploaded by SparsePrecond
// Hubs: h unknowns coupled to a large share of all others (a landmark ranged from every pose).  They are kept
// out of the dissection and handled by a Schur complement: with A = [A11 a; a^T alpha],
//     y1 = A11^-1 b1,   x2 = Sc^-1 (b2 - a^T y1),   x1 = y1 - U x2,   U = A11^-1 a,   Sc = alpha - a^T U,
// i.e. one sparse dot per hub after the level replay and a rank-h correction folded into the final permutation.
struct PartInvHub {
  int h = 0;
  std::vector<int> idx;       // original index of each hub unknown
  std::vector<int> ap;        // h + 1 offsets into apos / aval: column q of a
  std::vector<int> apos;      // where the entry's y1 value lives (buffer * k + permuted position)
  std::vector<double> aval;
  std::vector<double> U;      // k x h, row j = permuted position j
  std::vector<double> Sinv;   // h x h, row-major
};

struct PartInvHost {
  int k = 0;                     // unknowns handled by the level replay (hubs excluded)
  int kfull = 0;                 // size of the system
  PartInvHub hub;
  std::vector<SpLevel> levels;   // forward levels (leaves first), then backward levels (root first)
  int nforward = 0;
  std::vector<MWave> mwaves;   // the wave records of all launches
  std::vector<double> vals;
  // set by the caller before the build: the stored weights are streamed there while they are formed and `vals` stays
  // empty (the product: the 6.8 GB of the whole 100k lattice never exist on the host); null: weights in `vals`
  struct WeightSink *sink = nullptr;
  std::vector<MirrorRange> mirrors;  // device copies of the builder's sources (from the factor), for the sink
  bool sources_on_device_only = false;  // some sources are NOT readable on the host: only a device fill can form the weights
  long long nvals = 0;  // number of stored weights, wherever they went
  bool weights_ok = true;  // false: the sink refused them
  std::vector<int> idxs;
  std::vector<int> perm;         // permuted position -> original unknown
  std::vector<int> out_off;      // permuted position -> where its final value lives (buffer * k + position)
  long nnzL = 0;
  int npieces = 0;
  double weights_read_per_apply = 0;  // doubles of stored weights one solve streams
};

// A Cholesky factor P A P^T = L L^T handed over by dissection pieces: piece s holds the columns [c0, c0 + c) of L as
// a dense (c + m) x c panel (row-major): the lower-triangular diagonal block over the m rows listed in rows
// (permuted numbering, ascending; rows that are entirely zero are allowed).  Produced by the host factorisation
// (piecewise_from_chol) or by the device factorisation (device_chol.h).
struct PieceFactor {
  int c0 = 0, c = 0;
  std::vector<int> rows;
  std::vector<double> panel;
  // wide pieces may arrive already inverted by the device (device_chol.h): the panel then holds L11^-1 over
  // W = -L21 L11^-1, and for a piece without rows below Mtop = L11^-T L11^-1 (c x c, both triangles)
  bool inverted = false;
  std::vector<double> Mtop;
  // the device factorisation hands all panels over in ONE host block (PiecewiseFactor::block): views instead of 3 GB
  // of per-piece copies for the whole 100k lattice
  const double *panel_view = nullptr, *Mtop_view = nullptr;
  const double *pan() const { return panel_view ? panel_view : panel.data(); }
  const double *mtop() const { return Mtop_view ? Mtop_view : (Mtop.empty() ? nullptr : Mtop.data()); }
};
struct PiecewiseFactor {
  int n = 0, nhub = 0;
  long nnzL = 0;
  std::vector<int> perm, iperm;
  std::vector<PieceFactor> pieces;
  std::shared_ptr<void> block;  // keeps the storage behind the pieces' views alive
  // set by the caller BEFORE the factorisation: keep the panels and every piece's M on the device and describe them in
  // `mirrors` (the M then exist on the device only: Mtop_view is an address, not data)
  bool want_device_sources = false;
  std::vector<MirrorRange> mirrors;
  bool m_on_device_only = false;
};
void piecewise_from_chol(const SparseChol &chol, PiecewiseFactor *out);

// A symmetric positive definite (both triangles); block = unknowns ordered together.  false => not PD.
// The first form factorises on the host; the second takes a factor of A in the order nd_top_default() asks for.
bool build_partitioned_inverse(const HostCsr &A, int block, int nthreads, PartInvHost *out);
bool build_partitioned_inverse_from(const HostCsr &A, const PiecewiseFactor &F, int nthreads, PartInvHost *out);

// host reference of the device schedule (tests of the builder without a GPU): Z = R A^-1, R and Z are r x k
// column-major
void partitioned_inverse_apply_host(const PartInvHost &P, int r, const double *R, double *Z);

}  // namespace dcora
