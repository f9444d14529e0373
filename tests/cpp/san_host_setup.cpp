// Host-side set-up code of the product (ordering with the dense top, symbolic analysis of the device Cholesky and its
// host executor, partitioned-inverse builder with the deferred parallel weight fill, the host replay) run under
// AddressSanitizer + UndefinedBehaviorSanitizer (tests/test_host_sanitizers.py builds this file together with the three
// host sources; GPU sanitizers are not available on the pool, so the host half is what can be checked this way).
// argv: pairs of (matrix file written by the test: n, nnz, rowptr, colidx, values; block size).
#include "device_chol.h"
#include "sparse_precond.h"
#include <cstdio>
#include <cmath>
#include <random>
using namespace dcora;
static HostCsr load(const char*p){FILE*f=fopen(p,"rb"); int n,nnz; if(fread(&n,4,1,f)!=1||fread(&nnz,4,1,f)!=1) abort(); HostCsr A; A.n=A.ncols=n; A.rp.resize(n+1); A.ci.resize(nnz); A.v.resize(nnz);
 if(fread(A.rp.data(),4,n+1,f)!=(size_t)n+1||fread(A.ci.data(),4,nnz,f)!=(size_t)nnz||fread(A.v.data(),8,nnz,f)!=(size_t)nnz) abort(); fclose(f); return A;}
int main(int argc,char**argv){
  for(int a=1;a<argc;a+=2){
    HostCsr A=load(argv[a]); int block=atoi(argv[a+1]);
    for(int top: {0, 3072}){
      CholSymbolic S; chol_symbolic(A, block, &S, top);
      std::vector<double> F; bool ok = S.arena < 400000000LL ? chol_numeric_host(S, A.v.data(), &F) : true;
      printf("%s top %d: pieces %zu levels %d arena %lld ok %d\n", argv[a], top, S.pieces.size(), S.nlev, S.arena, (int)ok);
    }
    PartInvHost P; bool ok = build_partitioned_inverse(A, block, 4, &P);
    int r=3; std::vector<double> R((size_t)r*A.n), Z((size_t)r*A.n); std::mt19937 g(1); std::normal_distribution<double> nd; for(auto&x:R)x=nd(g);
    partitioned_inverse_apply_host(P, r, R.data(), Z.data());
    // residual A Z^T = R^T
    double worst=0; for(int i=0;i<A.n;++i) for(int t=0;t<r;++t){ double s=0; for(int p=A.rp[i];p<A.rp[i+1];++p) s+=A.v[p]*Z[(size_t)A.ci[p]*r+t]; worst=std::max(worst,std::fabs(s-R[(size_t)i*r+t])); }
    printf("%s partinv ok %d levels %zu hub %d resid %.3e\n", argv[a], (int)ok, P.levels.size(), P.hub.h, worst);
    // the same build with the stored weights streamed through a sink in small chunks of recycled (dirty) memory
    struct Sink : WeightSink {
      std::vector<double> all, buf; long long cap = 5000, expect = 0; bool ordered = true;
      bool begin(long long total) override { all.assign((size_t)total, -1.0); buf.assign((size_t)cap, 7.0); return true; }
      long long chunk_cap() const override { return cap; }
      double *acquire(long long m) override { if (m > cap) return nullptr; std::fill(buf.begin(), buf.end(), 7.0); return buf.data(); }
      bool commit(long long off, long long m) override { ordered = ordered && off == expect; expect = off + m; std::copy(buf.begin(), buf.begin() + m, all.begin() + off); return true; }
      bool end() override { return true; }
    } sink;
    PartInvHost Q; Q.sink = &sink; bool ok2 = build_partitioned_inverse(A, block, 3, &Q);
    long long differ = (long long)P.vals.size() != Q.nvals || !sink.ordered || sink.expect != Q.nvals;
    if (!differ) for (size_t i = 0; i < P.vals.size(); ++i) differ += P.vals[i] != sink.all[i];
    printf("%s streamed ok %d weights %lld differ %lld\n", argv[a], (int)ok2, Q.nvals, differ);
    if (differ) return 3;
  }
}
