/* Host-side AddressSanitizer driver for the C ABI: every entry point is called the way a careless
 * caller would -- empty batches, NULL pointers, out-of-range sizes -- and the return codes are checked; the library's
 * host code (argument checks, launch plumbing, error strings, the row-count helper) runs instrumented
 * (python -m bayesian_cbf_amd.build --asan).  Runs on a box WITHOUT a GPU: a call that gets as far as launching must
 * come back with BCBF_ELAUNCH and a message, not crash.  Built and run by tests/test_cabi_asan_cpu.py. */
#include <stdio.h>
#include <string.h>
#include "bcbf.h"

static int fails = 0;
#define EXPECT(call, want)                                                          \
    do {                                                                            \
        int rc_ = (call);                                                           \
        if (rc_ != (want)) { printf("FAIL %s -> %d (want %d)\n", #call, rc_, (want)); ++fails; } \
    } while (0)

int main(void) {
    double d[64] = {0};
    float f[64] = {0};
    int i[16] = {0};
    EXPECT(bcbf_version() >= 1 ? 0 : 1, 0);
    EXPECT(bcbf_lop_elems_f32(512) == (size_t)(512 * 514 / 2 + 32 * 512) ? 0 : 1, 0);
    EXPECT(bcbf_lop_elems_f64(1) == (size_t)(32 * 34 / 2 + 32 * 32) ? 0 : 1, 0);
    /* empty batches: every entry point returns OK without touching its pointers */
    EXPECT(bcbf_kb_build_f64(0, 0, 0, 0, 0, 0, 0, 0, 8, 2, 1, 0), BCBF_OK);
    EXPECT(bcbf_refit_f64(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 2, 1, 0), BCBF_OK);
    EXPECT(bcbf_refit_f32(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 2, 1, 0), BCBF_OK);
    EXPECT(bcbf_potrf_f64(0, 0, 0, 0, 0, 8, 0), BCBF_OK);
    EXPECT(bcbf_potrs_f32(0, 0, 0, 0, 0, 0, 0, 8, 2, 1, 0), BCBF_OK);
    EXPECT(bcbf_chol_append_f64(0, 0, 0, 0, 0, 0, 8, 0), BCBF_OK);
    EXPECT(bcbf_gp_append_f64(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 2, 1, 0), BCBF_OK);
    EXPECT(bcbf_potri_f64(0, 0, 0, 8, 0), BCBF_OK);
    EXPECT(bcbf_gp_reserve_f64(0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 0, 16, 2, 1, 0), BCBF_OK);
    EXPECT(bcbf_syrk_lt_f32(0, 0, 0, 8, 0), BCBF_OK);
    EXPECT(bcbf_mll_grad_f64(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 2, 1, 0, 0), BCBF_OK);
    EXPECT(bcbf_mll_grad_work_bytes(1, 512, 2) == (size_t)(8 * 32 * 26) ? 0 : 1, 0);
    EXPECT(bcbf_mll_grad_work_bytes(64, 512, 2) == (size_t)(8 * 64 * 2 * 26) ? 0 : 1, 0);       /* batches: one slot set per 256-row chunk (row form) */
    EXPECT(bcbf_kb_build_matern52_f32(0, 0, 0, 0, 0, 0, 0, 0, 8, 2, 1, 0), BCBF_OK);
    EXPECT(bcbf_posterior_query_matern52_f64(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 8, 3, 2, 0), BCBF_OK);
    EXPECT(bcbf_posterior_query_reserved_f32(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 16, 3, 2, 0), BCBF_OK);
    EXPECT(bcbf_gp_append_reserved_f64(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 16, 3, 2, 0), BCBF_OK);
    EXPECT(bcbf_gp_tail_step_f32(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 0, 8, 16, 16, 3, 2, 1, 0), BCBF_OK);
    EXPECT(bcbf_posterior_step_f32(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 3, 2, 0), BCBF_OK);
    EXPECT(bcbf_posterior_query_f64(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 8, 3, 2, 0), BCBF_OK);
    EXPECT(bcbf_posterior_shared_f32(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 3, 2, 0), BCBF_OK);
    EXPECT(bcbf_posterior_jets_f64(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 3, 2, 0), BCBF_OK);
    EXPECT(bcbf_cbc_terms_f64(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 3, 3, 2, 0), BCBF_OK);
    EXPECT(bcbf_socp_f32(0, 0, 0, 0, 0, 0, 0, 0, 0, 3, 2, 20, 0), BCBF_OK);
    EXPECT(bcbf_cbc_socp_f64(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 3, 3, 2, 20, 0), BCBF_OK);
    EXPECT(bcbf_coneqp_f64(0, 0, 0, 0, 3, 0, i, 1, 0, 0, 0, 0, 20, 0), BCBF_OK);
    EXPECT(bcbf_unicycle_constraints_f64(0, 0, 0, 0, 10.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 0, 2, 0), BCBF_OK);
    EXPECT(bcbf_unicycle_step_f32(0, 0, 0.1f, 1.0f, 0, 0), BCBF_OK);
    EXPECT(bcbf_rollout_stats_f64(0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 3, 0), BCBF_OK);
    EXPECT(bcbf_rollout_stats_f64(d, d, i, d, 0, d, d, i, 1, 2, 3, 0), BCBF_EINVAL);              /* obstacle rows without gammas */
    /* bad arguments with a non-empty batch: refused before anything is dereferenced or launched */
    EXPECT(bcbf_kb_build_f64(0, d, d, d, d, 0, d, 1, 8, 2, 1, 0), BCBF_EINVAL);                  /* X == NULL */
    EXPECT(bcbf_refit_f64(d, d, d, d, d, 0, 0, d, 0, i, 1, 8, 2, 1, 0), BCBF_EINVAL);            /* Lop == NULL */
    EXPECT(bcbf_refit_f32(f, f, f, f, f, 0, f, f, 0, i, 1, 8, 99, 1, 0), BCBF_EINVAL);           /* n out of range */
    EXPECT(bcbf_potrs_f64(d, d, d, d, d, 0, 1, 4096, 2, 1, 0), BCBF_EINVAL);                     /* N > 2048 */
    EXPECT(bcbf_posterior_step_f64(d, d, d, d, d, d, d, d, d, 0, d, d, 1, 8, 3, 7, 0), BCBF_EINVAL);   /* m out of range */
    EXPECT(bcbf_posterior_jets_f64(d, d, d, d, d, d, d, d, d, d, d, 0, d, 0, 0, 1, 8, 3, 2, 0), BCBF_EINVAL);  /* G == NULL */
    EXPECT(bcbf_socp_f64(d, d, d, d, d, d, i, i, 1, 5, 2, 20, 0), BCBF_EINVAL);                  /* K > 4 */
    EXPECT(bcbf_coneqp_f64(d, d, d, d, 99, 0, i, 1, d, i, i, 1, 20, 0), BCBF_EINVAL);            /* nv out of range */
    EXPECT(bcbf_chol_append_f64(d, d, d, d, i, 1, 32, 0), BCBF_EINVAL);                           /* in place across a padding boundary */
    EXPECT(bcbf_gp_reserve_f64(d, d, d, d, d + 1, d + 1, d + 1, d + 1, 1, 8, 0, 4, 2, 1, 0), BCBF_EINVAL);   /* capacity < N */
    EXPECT(bcbf_gp_reserve_f64(d, d, d, d, d, d, d, d, 1, 8, 0, 16, 2, 1, 0), BCBF_EINVAL);         /* in == out */
    EXPECT(bcbf_syrk_lt_f64(d, d, 1, 8, 0), BCBF_EINVAL);                                            /* in place */
    EXPECT(bcbf_gp_append_reserved_f64(d, d, d, d, d, d, d, d, d, d, d, 0, i, d, d, d, 0, 0, 0, 1, 16, 16, 3, 2, 0), BCBF_EINVAL);  /* full */
    EXPECT(bcbf_gp_append_reserved_f64(d, d, d, d, d, d, d, d, d, d, d, 0, i, d, d, d, d, 0, 0, 1, 8, 16, 3, 2, 0), BCBF_EINVAL);   /* xq without Mk */
    EXPECT(bcbf_posterior_query_reserved_f64(d, d, d, d, d, d, d, d, d, 0, d, d, 0, 1, 16, 8, 3, 2, 0), BCBF_EINVAL);     /* Ncap < N */
    EXPECT(bcbf_gp_tail_step_f64(d, d, d, d, d, d, d, d, d, d, d, d, 0, d, d, i, d, d, d, d, 0, 0, 0, 1, 8, 8, 8, 32, 32, 3, 2, 1, 0), BCBF_EINVAL);   /* the tail is full */
    EXPECT(bcbf_gp_tail_step_f64(d, d, d, d, d, d, d, d, d, d, d, d, 0, d, d, i, d, d, d, d, 0, 0, 0, 1, 8, 0, 65, 128, 128, 3, 2, 1, 0), BCBF_EINVAL); /* tcap > 64 */
    EXPECT(bcbf_gp_tail_step_f64(d, d, d, d, d, d, d, d, d, d, d, d, 0, d, d, i, d, d, d, d, 0, 0, 0, 1, 8, 0, 8, 32, 4, 3, 2, 1, 0), BCBF_EINVAL);    /* operator laid out for fewer points than it holds */
    EXPECT(bcbf_gp_tail_step_f64(d, d, d, d, d, d, d, d, d, d, d, d, 0, d, d, i, d, d, d, d, d, 0, 0, 1, 8, 0, 8, 32, 32, 3, 2, 1, 0), BCBF_EINVAL);   /* one raw store without the others */
    /* round 6: the batched fit, the Gram, the host-free retry, the observing control step, the tail commit */
    EXPECT(bcbf_fit_param_count(3, 2, 3, 3) == 3 + 1 + 9 + 3 + 9 + 3 + 9 ? 0 : 1, 0);
    EXPECT(bcbf_fit_param_count(3, 2, 1, 1) == 3 + 1 + 3 + 3 + 3 + 3 + 9 ? 0 : 1, 0);
    EXPECT(bcbf_fit_param_count(9, 2, 3, 3) < 0 ? 0 : 1, 0);                                      /* n beyond BCBF_MAX_STATE_DIM */
    EXPECT(bcbf_fit_derive_f64(0, 0, 0, 0, 0, 0, 0, 0, 0, 3, 2, 3, 3, 0), BCBF_OK);
    EXPECT(bcbf_fit_derive_f64(0, d, d, d, d, d, d, d, 1, 3, 2, 3, 3, 0), BCBF_EINVAL);           /* theta == NULL */
    EXPECT(bcbf_fit_derive_f32(f, f, f, f, f, f, 0, f, 1, 3, 2, 3, 3, 0), BCBF_EINVAL);           /* logdetA without Ainv */
    EXPECT(bcbf_fit_derive_f64(d, d, d, d, d, d, d, d, 1, 3, 2, 4, 3, 0), BCBF_EINVAL);           /* rank beyond n */
    EXPECT(bcbf_fit_adam_step_f64(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 3, 2, 3, 3, 1, 0.1, 0.9, 0.999, 1e-8, 0, 0), BCBF_OK);
    EXPECT(bcbf_fit_adam_step_f64(d, 0, 0, d, d, d, d, d, d, d, d, 0, d, 0, 1, 8, 3, 2, 3, 3, 1, 0.1, 0.9, 0.999, 1e-8, 0, 0), BCBF_EINVAL);   /* a step without moment buffers */
    EXPECT(bcbf_fit_adam_step_f64(d, d, d, d, d, d, d, d, d, d, d, 0, d, 0, 1, 8, 3, 2, 3, 3, 1, 0.1, 1.5, 0.999, 1e-8, 0, 0), BCBF_EINVAL);   /* beta1 out of range */
    { double prior[2] = {-1.0, 1.0};
      EXPECT(bcbf_fit_adam_step_f64(d, d, d, d, d, d, d, d, d, d, d, 0, d, 0, 1, 8, 3, 2, 3, 3, 0, 0.1, 0.9, 0.999, 1e-8, prior, 0), BCBF_EINVAL); } /* Gamma prior with a non-positive concentration */
    EXPECT(bcbf_kinv_apply_f64(0, 0, 0, 0, 8, 3, 0), BCBF_OK);
    EXPECT(bcbf_kinv_apply_f64(d, d, d, 1, 8, 3, 0), BCBF_EINVAL);                                /* in place */
    EXPECT(bcbf_kinv_apply_f32(f, f + 32, f + 48, 1, 4, 9, 0), BCBF_EINVAL);                      /* more than 8 columns */
    EXPECT(bcbf_gram_f32(0, 0, 0, 0, 0, 32, 2, 0), BCBF_OK);
    EXPECT(bcbf_gram_f64(d, d, d, 1, 1, 32, 2, 0), BCBF_EINVAL);                                  /* output aliases an input */
    EXPECT(bcbf_gram_f64(d, d, d + 32, 1, 1, 32, 13, 0), BCBF_EINVAL);                            /* C beyond BCBF_MAX_TASK_DIM */
    EXPECT(bcbf_predict_fullmat_f64(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 2, 1, 0, 0), BCBF_OK);
    EXPECT(bcbf_predict_fullmat_f64(d, d, d, d, d, d, d, d, d, d, 0, d, d, d, d, 0, 0, 2, 8, 2, 1, 0, 0), BCBF_EINVAL);   /* neither BkXX nor Kron */
    EXPECT(bcbf_predict_fullmat_f64(d, d, d, d, d, d, d, d, d, d, 0, d, d, d, d, d, 0, 2, 8, 2, 1, 7, 0), BCBF_EINVAL);   /* unknown kernel kind */
    EXPECT(bcbf_refit_retry_f64(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 2, 1, 0), BCBF_OK);
    EXPECT(bcbf_refit_retry_f64(d, d, d, d, d, d, d, d, i, i, 1, 8, 2, 1, 0), BCBF_EINVAL);      /* prev_info == info */
    EXPECT(bcbf_refit_retry_f32(f, f, f, f, f, f, f, f, 0, i, 1, 8, 2, 1, 0), BCBF_EINVAL);      /* no previous info */
    EXPECT(bcbf_gp_tail_commit_f32(0, 0, 0, 0, 64, 32, 32, 128, 0), BCBF_OK);
    EXPECT(bcbf_gp_tail_commit_f64(d, d, d, 1, 40, 32, 32, 128, 0), BCBF_EINVAL);                 /* N0 no multiple of 32 */
    EXPECT(bcbf_gp_tail_commit_f64(d, d, d, 1, 64, 31, 32, 128, 0), BCBF_EINVAL);                 /* a partial block */
    EXPECT(bcbf_gp_tail_commit_f64(d, d, d, 1, 64, 32, 32, 80, 0), BCBF_EINVAL);                  /* no room in the reservation */
    EXPECT(bcbf_unicycle_control_step_observe_f64(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 10.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                                  0.01, 1.0, 0, 8, 2, 20, 0, 0, 0, 0, 0, 1, 0, 3, 0, 0, 0), BCBF_OK);
    EXPECT(bcbf_unicycle_control_step_observe_f64(0, 0, 0, 0, 0, 0, 0, 0, d, d, d, d, d, 10.0, d, d, d, d, 1.0, d, d, d, d, d, d, d, d, d, d, d, d, i, d, i, i,
                                                  0.01, 1.0, 1, 0, 2, 20, 0, 0, d, 0, d, 1, 0, 1, 0, 0, 0), BCBF_EINVAL);   /* observation outputs: all three or none */
    EXPECT(bcbf_unicycle_control_step_observe_f64(0, 0, 0, 0, 0, 0, 0, 0, d, d, d, d, d, 10.0, d, d, d, d, 1.0, d, d, d, d, d, d, d, d, d, d, d, d, i, d, i, i,
                                                  0.0, 1.0, 1, 0, 2, 20, 0, 0, d, d, d, 1, 0, 1, 0, 0, 0), BCBF_EINVAL);    /* an observation needs dt > 0 */
    {   /* the row-count helper is pure host logic */
        int kinds[4] = {1, 2, 0, 1};
        EXPECT(bcbf_controller_cones_rows(kinds, 4, 2, 1) == 1 + 4 + 3 * 4 ? 0 : 1, 0);
        int bad[1] = {7};
        EXPECT(bcbf_controller_cones_rows(bad, 1, 2, 0) < 0 ? 0 : 1, 0);
    }
    {   /* a valid call on a machine without a GPU: the launch fails, the error is reported, nothing crashes */
        int rc = bcbf_unicycle_step_f64(d, d, 0.1, 1.0, 1, 0);
        const char* msg = bcbf_last_error();
        if (rc == BCBF_OK) printf("note: a GPU is present, launch succeeded\n");
        else if (rc != BCBF_ELAUNCH || msg == 0 || strlen(msg) == 0) { printf("FAIL launch path rc=%d msg=%s\n", rc, msg ? msg : "(null)"); ++fails; }
    }
    printf(fails ? "asan driver: %d FAILURES\n" : "asan driver: ok\n", fails);
    return fails ? 1 : 0;
}
