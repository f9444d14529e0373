// see precond_cache.h
#include "precond_cache.h"
#include "env.h"

#include <cstdlib>
#include <cstring>
#include <list>
#include <mutex>

namespace dcora {

namespace {

inline uint64_t mix(uint64_t h, uint64_t w) {
  h ^= w;
  h *= 0x9E3779B97F4A7C15ull;
  h ^= h >> 29;
  return h;
}
void hash_bytes(const void *p, size_t n, uint64_t *h0, uint64_t *h1) {
  const unsigned char *b = (const unsigned char *)p;
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t w;
    std::memcpy(&w, b + i, 8);
    *h0 = mix(*h0, w);
    *h1 = mix(*h1 + 0xD1B54A32D192ED03ull, w ^ (*h0 >> 7));
  }
  uint64_t w = 0;
  if (i < n) std::memcpy(&w, b + i, n - i);
  *h0 = mix(*h0, w ^ n);
  *h1 = mix(*h1, w + n);
}

struct Slot {
  PrecondKey key;
  PrecondEntry e;
};
std::mutex g_mu;
std::list<Slot> g_lru;  // front = most recently used
size_t g_bytes = 0;
double g_hits = 0, g_misses = 0;

size_t budget() {
  static const size_t b = [] {
    const double mb = env::precond_cache_mb();
    return (size_t)(mb * 1024.0 * 1024.0);
  }();
  return b;
}

}  // namespace

PrecondKey make_precond_key(const HostCsr &Q, double reg, int block, int device, bool sparse) {
  PrecondKey k;
  k.k = Q.n;
  k.nnz = (int)Q.ci.size();
  k.block = block;
  k.device = device;
  k.sparse = sparse ? 1 : 0;
  k.reg = reg;
  k.h0 = 0x243F6A8885A308D3ull;
  k.h1 = 0x13198A2E03707344ull;
  hash_bytes(Q.rp.data(), Q.rp.size() * sizeof(int), &k.h0, &k.h1);
  hash_bytes(Q.ci.data(), Q.ci.size() * sizeof(int), &k.h0, &k.h1);
  hash_bytes(Q.v.data(), Q.v.size() * sizeof(double), &k.h0, &k.h1);
  const size_t nz = Q.ci.size(), nr = Q.rp.size();
  for (size_t i = 0; i < nz; ++i) {
    k.vsum += Q.v[i];
    k.vabs += Q.v[i] < 0 ? -Q.v[i] : Q.v[i];
    k.cisum += Q.ci[i];
  }
  for (int q = 0; q < PrecondKey::kSamples; ++q) {
    if (nz) {
      const size_t i = (size_t)((double)q / PrecondKey::kSamples * (double)nz);
      k.vsample[q] = Q.v[i];
      k.cisample[q] = Q.ci[i];
    }
    if (nr) k.rpsample[q] = Q.rp[(size_t)((double)q / PrecondKey::kSamples * (double)nr)];
  }
  return k;
}

bool precond_cache_find(const PrecondKey &key, PrecondEntry *out, bool count) {
  if (budget() == 0) return false;
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto it = g_lru.begin(); it != g_lru.end(); ++it)
    if (it->key == key) {
      g_lru.splice(g_lru.begin(), g_lru, it);
      *out = g_lru.front().e;
      if (count) {
        if (g_lru.front().e.prebuilt) {
          g_lru.front().e.prebuilt = false;  // claimed by the problem it was built for
          g_misses += 1;
        } else {
          g_hits += 1;
        }
      }
      return true;
    }
  if (count) g_misses += 1;
  return false;
}

void precond_cache_insert(const PrecondKey &key, const PrecondEntry &e) {
  if (budget() == 0 || e.bytes > budget()) return;
  std::lock_guard<std::mutex> lk(g_mu);
  for (const Slot &s : g_lru)
    if (s.key == key) return;
  g_lru.push_front(Slot{key, e});
  g_bytes += e.bytes;
  while (g_bytes > budget() && g_lru.size() > 1) {
    g_bytes -= g_lru.back().e.bytes;
    g_lru.pop_back();  // device memory is released when the last problem using the image goes away
  }
}

void precond_cache_stats(double *s) {
  std::lock_guard<std::mutex> lk(g_mu);
  s[0] = g_hits;
  s[1] = g_misses;
  s[2] = (double)g_lru.size();
  s[3] = (double)g_bytes;
}

void precond_cache_clear() {
  std::lock_guard<std::mutex> lk(g_mu);
  g_lru.clear();
  g_bytes = 0;
}

}  // namespace dcora
