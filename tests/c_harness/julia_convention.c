/*
 * TEST INFRASTRUCTURE -- plain-C caller of libproxsdp_hip.so that does exactly what the Julia shim
 * (julia/ProxSDPHip.jl) does at /root/reference/src/MOI_wrapper.jl:310, without Julia:
 *
 *   - options live in an OPAQUE 1024-byte buffer (ProxSDPHip.jl `OPTIONS_BYTES`), filled by
 *     proxsdp_hip_default_options and then BY NAME through proxsdp_hip_set_option, the way
 *     MOI.RawOptimizerAttribute reflection does (MOI_wrapper.jl:84-93); the C struct layout is never
 *     touched from this file;
 *   - the problem is the reference's `sdp_wiki` known-answer test (test/moi_proxsdp_unit.jl:302-338:
 *     3x3 PSD, X1 = X3 = X6 = 1, -0.2 <= X2 <= -0.1, 0.4 <= X5 <= 0.5, min X4 -> -0.978) in the layout
 *     Julia hands over: SparseMatrixCSC{Float64,Int64} with 1-BASED colptr / rowval
 *     (structs.jl:36-37), SDPSet.vec_i 1-based (structs.jl:44-48), index_base = 1, psd_ptr 0-based
 *     offsets;
 *   - results land in caller-allocated arrays, errors come back as codes + proxsdp_hip_last_error().
 *
 * Modes:  `prep`  -- host-only: options by name + proxsdp_host_preprocess (no GPU needed; CPU test suite)
 *         `solve` -- proxsdp_hip_solve on device 0 (GPU test suite)
 * Output: key=value lines parsed by tests/test_julia_convention.py.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "proxsdp_hip.h"

#define OPTIONS_BYTES 1024

static int set_by_name(uint64_t* buf, const char* name, double v) {
    int rc = proxsdp_hip_set_option((proxsdp_options*)buf, name, v);
    if (rc != 0) fprintf(stderr, "No parameter matching %s (rc %d: %s)\n", name, rc, proxsdp_hip_last_error());
    return rc;
}

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "prep";
    /* ---- options: opaque buffer, by name */
    uint64_t optbuf[OPTIONS_BYTES / 8];
    memset(optbuf, 0, sizeof(optbuf));
    proxsdp_hip_default_options((proxsdp_options*)optbuf);
    const int64_t struct_size = (int64_t)optbuf[0];          /* first field, by contract */
    printf("abi_version=%d\n", proxsdp_hip_abi_version());
    printf("options_struct_size=%lld\n", (long long)struct_size);
    if (struct_size <= 0 || struct_size > OPTIONS_BYTES) {
        fprintf(stderr, "proxsdp_options (%lld bytes) does not fit the shim's %d-byte buffer\n",
                (long long)struct_size, OPTIONS_BYTES);
        return 2;
    }
    if (set_by_name(optbuf, "tol_gap", 1e-6) || set_by_name(optbuf, "tol_feasibility", 1e-6) ||
        set_by_name(optbuf, "log_verbose", 0.0) || set_by_name(optbuf, "time_limit", 60.0))
        return 3;
    double back = 0.0;
    if (proxsdp_hip_get_option((const proxsdp_options*)optbuf, "tol_gap", &back) != 0) return 3;
    printf("tol_gap=%.17g\n", back);
    if (proxsdp_hip_set_option((proxsdp_options*)optbuf, "unsupportedarg", 10.0) == 0) {
        fprintf(stderr, "unknown option name was accepted\n");
        return 3;
    }
    printf("unknown_option_error=%s\n", proxsdp_hip_last_error());

    /* ---- sdp_wiki in Julia's layout (1-based).  Variables: MOI triangle order X11 X12 X22 X13 X23 X33.
     * A (3 x 6): rows select X1, X3, X6.  Column-compressed: col1 -> row1, col3 -> row2, col6 -> row3. */
    int64_t A_colptr[7] = {1, 2, 2, 3, 3, 3, 4};
    int64_t A_rowval[3] = {1, 2, 3};
    double A_nzval[3] = {1.0, 1.0, 1.0};
    /* G (4 x 6): X2 <= -0.1, -X2 <= 0.2, X5 <= 0.5, -X5 <= -0.4 */
    int64_t G_colptr[7] = {1, 1, 3, 3, 3, 5, 5};
    int64_t G_rowval[4] = {1, 2, 3, 4};
    double G_nzval[4] = {1.0, -1.0, 1.0, -1.0};
    double b[3] = {1.0, 1.0, 1.0};
    double h[4] = {-0.1, 0.2, 0.5, -0.4};
    double c[6] = {0.0, 0.0, 0.0, 1.0, 0.0, 0.0};           /* min X4 (= 2 X13 on the triangle variable) */
    int64_t psd_ptr[2] = {0, 6};
    int64_t psd_idx[6] = {1, 2, 3, 4, 5, 6};
    int64_t soc_ptr[1] = {0};

    proxsdp_problem prob;
    memset(&prob, 0, sizeof(prob));
    prob.n = 6; prob.p = 3; prob.m = 4;
    prob.A.nrows = 3; prob.A.ncols = 6; prob.A.colptr = A_colptr; prob.A.rowval = A_rowval; prob.A.nzval = A_nzval;
    prob.G.nrows = 4; prob.G.ncols = 6; prob.G.colptr = G_colptr; prob.G.rowval = G_rowval; prob.G.nzval = G_nzval;
    prob.b = b; prob.h = h; prob.c = c;
    prob.n_psd = 1; prob.psd_ptr = psd_ptr; prob.psd_idx = psd_idx;
    prob.n_soc = 0; prob.soc_ptr = soc_ptr; prob.soc_idx = NULL;
    prob.index_base = 1;

    if (strcmp(mode, "prep") == 0) {
        int64_t order[6], inv[6];
        double cs[6], fro = 0.0;
        int rc = proxsdp_host_preprocess(&prob, order, inv, cs, &fro);
        if (rc != 0) { fprintf(stderr, "preprocess failed: %d %s\n", rc, proxsdp_hip_last_error()); return 4; }
        printf("order=");
        for (int k = 0; k < 6; ++k) printf("%lld%s", (long long)order[k], k < 5 ? "," : "\n");
        printf("c_scaled=");
        for (int k = 0; k < 6; ++k) printf("%.17g%s", cs[k], k < 5 ? "," : "\n");
        printf("frobenius=%.17g\n", fro);
        /* a 0-based caller error must be caught, not read out of bounds: colptr[0] != index_base */
        prob.index_base = 0;
        rc = proxsdp_host_preprocess(&prob, order, inv, cs, &fro);
        printf("wrong_base_rc=%d\n", rc);
        return rc == PROXSDP_E_INVALID ? 0 : 5;
    }

    double primal[6], dual_cone[6], dual_eq[3], dual_in[4], slack_eq[3], slack_in[4];
    proxsdp_result res;
    memset(&res, 0, sizeof(res));
    res.primal = primal; res.dual_cone = dual_cone; res.dual_eq = dual_eq; res.dual_in = dual_in;
    res.slack_eq = slack_eq; res.slack_in = slack_in; res.trace = NULL;
    int rc = proxsdp_hip_solve(&prob, (const proxsdp_options*)optbuf, &res);
    if (rc != 0) { fprintf(stderr, "solve failed: %d %s\n", rc, proxsdp_hip_last_error()); return 10 - rc; }
    printf("status=%d\n", res.status);
    printf("status_string=%s\n", res.status_string);
    printf("iter=%lld\n", (long long)res.iter);
    printf("objval=%.17g\n", res.objval);
    printf("dual_objval=%.17g\n", res.dual_objval);
    printf("primal=");
    for (int k = 0; k < 6; ++k) printf("%.17g%s", primal[k], k < 5 ? "," : "\n");
    printf("dual_eq=");
    for (int k = 0; k < 3; ++k) printf("%.17g%s", dual_eq[k], k < 2 ? "," : "\n");
    printf("dual_in=");
    for (int k = 0; k < 4; ++k) printf("%.17g%s", dual_in[k], k < 3 ? "," : "\n");
    printf("primal_feasible=%d\n", res.primal_feasible_user_tol);
    printf("dual_feasible=%d\n", res.dual_feasible_user_tol);
    return 0;
}
