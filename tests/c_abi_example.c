/* Plain-C consumer of include/ecne.h: proves the boundary is a C ABI (no C++ or torch types).
 * Loads a .r1cs, prints readR1CS's tuple, abstracts nothing, and reports what ecne_solve says when no
 * GPU is visible (ECNE_ENODEVICE) or the verdict when one is.  Built and run by tests/test_c_abi.py. */
#include <stdio.h>
#include "ecne.h"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    ecne_r1cs* f = NULL;
    int st = ecne_r1cs_load(argv[1], &f);
    if (st != ECNE_OK) { printf("load: %s\n", ecne_strerror(st)); return 1; }
    ecne_info info;
    ecne_r1cs_info(f, &info);
    const int64_t *known, *targets;
    size_t nk, nt;
    ecne_r1cs_io(f, &known, &nk, &targets, &nt);
    printf("constraints=%u nVars=%lld known=%zu targets=%zu nnz=%llu/%llu/%llu\n", info.n_constraints,
           (long long)info.n_vars, nk, nt, (unsigned long long)info.nnz[0], (unsigned long long)info.nnz[1],
           (unsigned long long)info.nnz[2]);
    ecne_system* sys = NULL;
    ecne_system_from_r1cs(f, &sys);
    ecne_opts opts = {0, 0, 0, 0, NULL};
    ecne_result* res = NULL;
    st = ecne_solve(sys, &opts, &res);
    if (st == ECNE_OK) {
        ecne_summary s;
        ecne_result_summary(res, &s);
        printf("solve: status=%d function_good=%d unique=%lld/%lld targets=%lld/%lld\n", s.status, s.function_good,
               (long long)s.unique_nontrivial, (long long)s.n_nontrivial, (long long)s.unique_targets, (long long)s.n_targets);
        ecne_result_free(res);
    } else {
        printf("solve: %s\n", ecne_strerror(st));
    }
    ecne_system_free(sys);
    ecne_r1cs_free(f);
    return 0;
}
