// ORACLE -- test infrastructure only (see oracle.hpp).
// Dense/sparse helpers and the sparse Cholesky that stands in for CHOLMOD
// (ref: src/Graph.cpp:1901-1917 preconditioner; src/DCORA_utils.cpp:1737-1747
// PSD test).  Fill-reducing ordering: minimum degree on the block-compressed
// graph; numeric phase: up-looking LL^T driven by the elimination tree.
#include <vector>
#include <algorithm>
#include <numeric>
#include <queue>
#include <set>
#include <unordered_set>

#include "oracle.hpp"

namespace orc {

double dot(const Mat &A, const Mat &B) {
  double s = 0;
  const size_t N = A.a.size();
  for (size_t i = 0; i < N; ++i) s += A.a[i] * B.a[i];
  return s;
}
double norm(const Mat &A) { return std::sqrt(dot(A, A)); }
void axpy(double a, const Mat &X, Mat &Y) {
  const size_t N = X.a.size();
  for (size_t i = 0; i < N; ++i) Y.a[i] += a * X.a[i];
}

// Y = X * Q.  Q is symmetric so column j of X*Q is sum_c Q(j,c) X(:,c)
// (ref: src/QuadraticProblem.cpp:42,58,67 dense x row-major-sparse product).
void spmm_right(const Mat &X, const CSR &Q, Mat &Y) {
  const int r = X.rows;
  if (Y.rows != X.rows || Y.cols != X.cols) Y = Mat(X.rows, X.cols);
  std::vector<double> accv((size_t)std::max(r, 1));
  double *acc = accv.data();
  for (int j = 0; j < Q.n; ++j) {
    for (int i = 0; i < r; ++i) acc[i] = 0;
    for (int p = Q.rp[j]; p < Q.rp[j + 1]; ++p) {
      const double q = Q.v[p];
      const double *xc = X.col(Q.ci[p]);
      for (int i = 0; i < r; ++i) acc[i] += q * xc[i];
    }
    double *yc = Y.col(j);
    for (int i = 0; i < r; ++i) yc[i] = acc[i];
  }
}

void spmv(const CSR &S, const double *x, double *y) {
  for (int i = 0; i < S.n; ++i) {
    double s = 0;
    for (int p = S.rp[i]; p < S.rp[i + 1]; ++p) s += S.v[p] * x[S.ci[p]];
    y[i] = s;
  }
}

CSR csr_from_triplets(int n, std::vector<int> &I, std::vector<int> &J,
                      std::vector<double> &V) {
  // counting sort by row, then sort/merge inside rows (duplicates summed)
  CSR A;
  A.n = n;
  A.rp.assign(n + 1, 0);
  const size_t m = I.size();
  for (size_t e = 0; e < m; ++e) A.rp[I[e] + 1]++;
  for (int i = 0; i < n; ++i) A.rp[i + 1] += A.rp[i];
  std::vector<int> pos(A.rp.begin(), A.rp.end() - 1), cj(m);
  std::vector<double> cv(m);
  for (size_t e = 0; e < m; ++e) {
    int p = pos[I[e]]++;
    cj[p] = J[e];
    cv[p] = V[e];
  }
  std::vector<int> rp2(n + 1, 0);
  std::vector<std::pair<int, double>> row;
  for (int i = 0; i < n; ++i) {
    row.clear();
    for (int p = A.rp[i]; p < A.rp[i + 1]; ++p) row.emplace_back(cj[p], cv[p]);
    std::sort(row.begin(), row.end(),
              [](const auto &a, const auto &b) { return a.first < b.first; });
    size_t q = 0;
    while (q < row.size()) {
      int c = row[q].first;
      double s = 0;
      while (q < row.size() && row[q].first == c) s += row[q++].second;
      A.ci.push_back(c);
      A.v.push_back(s);
    }
    rp2[i + 1] = (int)A.ci.size();
  }
  A.rp = rp2;
  return A;
}

CSR csr_add_diag(const CSR &A, double s) {
  std::vector<int> I, J;
  std::vector<double> V;
  I.reserve(A.nnz() + A.n);
  J.reserve(A.nnz() + A.n);
  V.reserve(A.nnz() + A.n);
  for (int i = 0; i < A.n; ++i) {
    for (int p = A.rp[i]; p < A.rp[i + 1]; ++p) {
      I.push_back(i);
      J.push_back(A.ci[p]);
      V.push_back(A.v[p]);
    }
    I.push_back(i);
    J.push_back(i);
    V.push_back(s);
  }
  return csr_from_triplets(A.n, I, J, V);
}

// Minimum-degree ordering on the graph compressed by `block` consecutive
// indices (poses), explicit elimination graph with lazy priority queue.
// Indices beyond the last full block are treated as singleton nodes.
std::vector<int> min_degree_order(const CSR &A, int block) {
  const int n = A.n;
  if (block < 1) block = 1;
  const int nb = (n + block - 1) / block;
  std::vector<std::unordered_set<int>> adj(nb);
  for (int i = 0; i < n; ++i)
    for (int p = A.rp[i]; p < A.rp[i + 1]; ++p) {
      int a = i / block, b = A.ci[p] / block;
      if (a != b) {
        adj[a].insert(b);
        adj[b].insert(a);
      }
    }
  using Item = std::pair<int, int>;  // (degree, node)
  std::priority_queue<Item, std::vector<Item>, std::greater<Item>> pq;
  for (int i = 0; i < nb; ++i) pq.emplace((int)adj[i].size(), i);
  std::vector<char> done(nb, 0);
  std::vector<int> border;
  std::vector<int> bperm;
  bperm.reserve(nb);
  while (!pq.empty()) {
    auto [deg, u] = pq.top();
    pq.pop();
    if (done[u] || deg != (int)adj[u].size()) continue;
    done[u] = 1;
    bperm.push_back(u);
    border.assign(adj[u].begin(), adj[u].end());
    for (int v : border) adj[v].erase(u);
    for (size_t a = 0; a < border.size(); ++a)
      for (size_t b = a + 1; b < border.size(); ++b) {
        adj[border[a]].insert(border[b]);
        adj[border[b]].insert(border[a]);
      }
    for (int v : border) pq.emplace((int)adj[v].size(), v);
    std::unordered_set<int>().swap(adj[u]);
  }
  std::vector<int> perm;
  perm.reserve(n);
  for (int bnode : bperm)
    for (int t = 0; t < block; ++t) {
      int idx = bnode * block + t;
      if (idx < n) perm.push_back(idx);
    }
  return perm;
}

bool Chol::factor(const CSR &A, int block) {
  n = A.n;
  ok = false;
  perm = min_degree_order(A, block);
  iperm.assign(n, 0);
  for (int i = 0; i < n; ++i) iperm[perm[i]] = i;
  // Upper triangle of the permuted matrix in CSC == rows of lower triangle:
  // for column j (new index) we need entries (i, j) with i <= j.
  std::vector<int> Cp(n + 1, 0), Ci;
  std::vector<double> Cx;
  {
    std::vector<std::vector<std::pair<int, double>>> cols(n);
    for (int io = 0; io < n; ++io) {
      int i = iperm[io];
      for (int p = A.rp[io]; p < A.rp[io + 1]; ++p) {
        int j = iperm[A.ci[p]];
        if (i <= j) cols[j].emplace_back(i, A.v[p]);
      }
    }
    for (int j = 0; j < n; ++j) {
      std::sort(cols[j].begin(), cols[j].end());
      for (auto &e : cols[j]) {
        Ci.push_back(e.first);
        Cx.push_back(e.second);
      }
      Cp[j + 1] = (int)Ci.size();
    }
  }
  // elimination tree (Liu) and column counts via row-subtree traversal
  std::vector<int> parent(n, -1), anc(n, -1);
  for (int k = 0; k < n; ++k)
    for (int p = Cp[k]; p < Cp[k + 1]; ++p) {
      int i = Ci[p];
      while (i != -1 && i < k) {
        int inext = anc[i];
        anc[i] = k;
        if (inext == -1) parent[i] = k;
        i = inext;
      }
    }
  std::vector<int> cnt(n, 1), mark(n, -1);
  for (int k = 0; k < n; ++k) {
    mark[k] = k;
    for (int p = Cp[k]; p < Cp[k + 1]; ++p) {
      int i = Ci[p];
      while (i < k && mark[i] != k) {
        cnt[i]++;
        mark[i] = k;
        i = parent[i];
      }
    }
  }
  Lp.assign(n + 1, 0);
  for (int j = 0; j < n; ++j) Lp[j + 1] = Lp[j] + cnt[j];
  Li.assign(Lp[n], 0);
  Lx.assign(Lp[n], 0.0);
  std::vector<int> fill(Lp.begin(), Lp.end() - 1);
  std::vector<double> x(n, 0.0);
  std::vector<int> stack(n), pattern(n);
  std::fill(mark.begin(), mark.end(), -1);
  // up-looking: row k of L from the sparse triangular solve L(0:k-1,0:k-1) y = A(0:k-1,k)
  for (int k = 0; k < n; ++k) {
    int top = n;
    mark[k] = k;
    double dk = 0;
    for (int p = Cp[k]; p < Cp[k + 1]; ++p) {
      int i = Ci[p];
      if (i == k) {
        dk += Cx[p];
        continue;
      }
      x[i] += Cx[p];
      int len = 0;
      while (mark[i] != k) {
        stack[len++] = i;
        mark[i] = k;
        i = parent[i];
      }
      while (len > 0) pattern[--top] = stack[--len];
    }
    for (; top < n; ++top) {
      int i = pattern[top];
      double lki = x[i] / Lx[Lp[i]];
      x[i] = 0;
      for (int p = Lp[i] + 1; p < fill[i]; ++p) x[Li[p]] -= Lx[p] * lki;
      dk -= lki * lki;
      int p = fill[i]++;
      Li[p] = k;
      Lx[p] = lki;
    }
    if (!(dk > 0)) return false;  // not positive definite (quick return)
    int p = fill[k]++;
    Li[p] = k;
    Lx[p] = std::sqrt(dk);
  }
  ok = true;
  return true;
}

void Chol::solve_vec(const double *b, double *xout) const {
  std::vector<double> y(n);
  for (int i = 0; i < n; ++i) y[i] = b[perm[i]];
  for (int j = 0; j < n; ++j) {
    y[j] /= Lx[Lp[j]];
    const double yj = y[j];
    for (int p = Lp[j] + 1; p < Lp[j + 1]; ++p) y[Li[p]] -= Lx[p] * yj;
  }
  for (int j = n - 1; j >= 0; --j) {
    double s = y[j];
    for (int p = Lp[j] + 1; p < Lp[j + 1]; ++p) s -= Lx[p] * y[Li[p]];
    y[j] = s / Lx[Lp[j]];
  }
  for (int i = 0; i < n; ++i) xout[perm[i]] = y[i];
}

// Z = V A^{-1}: the r rows of V are the right-hand sides
// (ref: src/QuadraticProblem.cpp:79 solve(INVEC.transpose()).transpose()).
void Chol::solve_rows(const Mat &V, Mat &Z) const {
  const int r = V.rows;
  if (Z.rows != V.rows || Z.cols != V.cols) Z = Mat(V.rows, V.cols);
  // permuted copy, r contiguous per unknown
  std::vector<double> y((size_t)n * r);
  for (int i = 0; i < n; ++i) {
    const double *src = V.col(perm[i]);
    for (int t = 0; t < r; ++t) y[(size_t)i * r + t] = src[t];
  }
  for (int j = 0; j < n; ++j) {
    double *yj = &y[(size_t)j * r];
    const double inv = 1.0 / Lx[Lp[j]];
    for (int t = 0; t < r; ++t) yj[t] *= inv;
    for (int p = Lp[j] + 1; p < Lp[j + 1]; ++p) {
      double *yi = &y[(size_t)Li[p] * r];
      const double l = Lx[p];
      for (int t = 0; t < r; ++t) yi[t] -= l * yj[t];
    }
  }
  for (int j = n - 1; j >= 0; --j) {
    double *yj = &y[(size_t)j * r];
    for (int p = Lp[j] + 1; p < Lp[j + 1]; ++p) {
      const double *yi = &y[(size_t)Li[p] * r];
      const double l = Lx[p];
      for (int t = 0; t < r; ++t) yj[t] -= l * yi[t];
    }
    const double inv = 1.0 / Lx[Lp[j]];
    for (int t = 0; t < r; ++t) yj[t] *= inv;
  }
  for (int i = 0; i < n; ++i) {
    double *dst = Z.col(perm[i]);
    for (int t = 0; t < r; ++t) dst[t] = y[(size_t)i * r + t];
  }
}

}  // namespace orc
