#include <cstdio>
#include <cstdlib>
#include "hostutil.h"
using namespace zk;
template <class F> int run(const char* name) {
    int bad = 0;
    F x = F::r2();
    for (int i = 0; i < 20000; i++) {
        x = fe_mul(fe_add(x, F::one()), fe_add(x, x));
        x.v[i & 7] ^= (uint32_t)rand();
        reduce_once(x);
        // make canonical: multiply by one
        x = fe_mul(x, F::r2());
        F a = fe_inv(x), b = fe_inv_fast(x);
        if (!(a == b)) bad++;
        if (!(fe_mul(b, x) == F::one()) && !x.is_zero()) bad++;
    }
    F z = F::zero();
    if (!(fe_inv_fast(z) == fe_inv(z))) bad++;
    F o = F::one();
    if (!(fe_inv_fast(o) == F::one())) bad++;
    printf("%s: %d mismatches\n", name, bad);
    return bad;
}
int main() {
    int bad = run<Fr>("Fr") + run<Fq>("Fq");
    // omega table
    for (uint32_t k = 0; k <= 28; k++) {
        Fr w = fr_root_of_unity_2_28();
        for (uint32_t i = k; i < 28; i++) w = fe_sqr(w);
        if (!(w == fr_omega(k))) bad++;
    }
    printf("total bad %d\n", bad);
    return bad ? 1 : 0;
}
