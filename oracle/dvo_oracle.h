/*
 * dvo_oracle.h -- C interface of the CPU ORACLE for the dense RGB-D alignment hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so, and only
 * as the checker / timed CPU baseline.  The product (libdvo_hip.so) never links or calls it.
 *
 * What it restates (all paths relative to /root/reference):
 *   driver          dvo_core/src/dense_tracking.cpp:131-376
 *   residual pass   dvo_core/src/dense_tracking_impl.cpp:122-393   (SSE variant = what runs)
 *   weights         dvo_core/src/dense_tracking_impl.cpp:640-707
 *   scale           dvo_core/src/dense_tracking_impl.cpp:566-638
 *   log-likelihood  dvo_core/src/dense_tracking_impl.cpp:406-425
 *   Jacobians       dvo_core/src/dense_tracking.cpp:448-476
 *   normal eq.      dvo_core/src/core/least_squares.cpp:35-80, math_sse.cpp:82-207
 *   selection       dvo_core/src/core/point_selection.cpp:68-152, point_selection.h:49-67
 *   image model     dvo_core/src/core/rgbd_image.cpp:38-55,127-172,186-204,245-296,419-500,534-543
 *   intrinsics      dvo_core/src/core/intrinsic_matrix.cpp:47-93
 *   depth ingest    dvo_core/src/core/surface_pyramid.cpp:65-105
 *   SE(3) exp/log   Sophus (un-vendored, unpinned: sophus/Makefile:5-8) -- closed forms restated
 *
 * PARITY STATUS.  The reference ships no tests, golden vectors or fixtures for this path (SURVEY.md
 * section 4 / 8c) and cannot be built with its own build system here (needs ROS, Eigen, OpenCV, Sophus,
 * TBB, boost).  The REF_SSE mode IS PINNED against the reference's own code: twelve dvo_core translation
 * units (driver, SSE passes, normal equations, selection, image model, intrinsics, depth ingest) are
 * compiled unmodified from /root/reference against stand-in headers for those libraries (oracle/shim/,
 * oracle/ref_bridge.cpp -> oracle/_ref/libdvo_ref.so), and tests/test_oracle_ref.py demands bit-identical
 * outputs pass by pass, for the image model, and for whole DenseTracker::match() runs.  SE(3) exp/log and
 * the 6x6 solve are external dependencies of the reference (Sophus, Eigen: un-vendored, unpinned) -- the
 * stand-ins use this oracle's closed forms for them, which are pinned against scipy instead.  The MATH mode
 * (below) is REF_SSE minus its order-/ISA-dependent quirks and cannot be pinned against reference code.
 *
 * Two semantic modes:
 *   DVO_ORACLE_REF_SSE  quirk-faithful: approximate reciprocal (_mm_rcp_ps) in projection and
 *                       weights, MXCSR round-toward-zero inside the residual loop, odd-N drop,
 *                       the computeScaleSse pairing bug, LL dropping n mod 50 terms, float
 *                       sequential accumulation, no FMA (2013-era SSE3 build).
 *   DVO_ORACLE_MATH     the same algorithm with the order-/ISA-dependent quirks removed
 *                       (exact division, round-to-nearest, all points, correct covariance,
 *                       full LL sum, float64 accumulators).  This is the semantics the GPU
 *                       implements; REF_SSE-vs-MATH deltas are reported by the tests.
 */
#ifndef DVO_ORACLE_H_
#define DVO_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { DVO_ORACLE_REF_SSE = 0, DVO_ORACLE_MATH = 1 };

/* Quirk by quirk: mode = DVO_ORACLE_QUIRKS | any of the bits below selects exactly those order-/ISA-dependent behaviours of
 * the reference on top of the MATH semantics (SURVEY.md section 8a quirk list).  DVO_ORACLE_QUIRKS | DVO_ORACLE_Q_ALL is
 * REF_SSE and DVO_ORACLE_QUIRKS alone is MATH, bit for bit (tests/test_oracle.py) -- so the distance between the reference's
 * trajectory and the exact arithmetic's can be attributed (tests/golden/make_quirk_table.py). */
enum {
  DVO_ORACLE_QUIRKS = 0x100,
  DVO_ORACLE_Q_RCP_PROJECTION = 1,     /* Q1: u = x * _mm_rcp_ps(z)            dense_tracking_impl.cpp:192 */
  DVO_ORACLE_Q_RCP_WEIGHTS = 2,        /* Q1: w = 7 * _mm_rcp_ps(5 + r^T P r)  dense_tracking_impl.cpp:700 */
  DVO_ORACLE_Q_ROUND_TOWARD_ZERO = 4,  /* Q2: MXCSR RZ inside the residual loop  dense_tracking_impl.cpp:165-167 */
  DVO_ORACLE_Q_DROP_ODD = 8,           /* Q3: odd trailing point ignored         dense_tracking_impl.cpp:169 */
  DVO_ORACLE_Q_SCALE_PAIRING = 16,     /* Q6: first residual of each pair used twice, float accumulation  :614-615 */
  DVO_ORACLE_Q_LOGLIK_TAIL = 32,       /* Q7: n mod 50 terms dropped, products of 50                      :406-425 */
  DVO_ORACLE_Q_FLOAT_NORMAL_EQ = 64,   /* float sequential 2x2-blocked accumulation of A, b               math_sse.cpp:82-178 */
  DVO_ORACLE_Q_ALL = 127,
  /* experiment, not part of REF_SSE: the pairing of Q6 as a formula over the compaction ranks, accumulated in float64 */
  DVO_ORACLE_X_PAIRING_F64 = 0x200
};

/* termination criteria, same numbering as dense_tracking.h:71-81 */
enum {
  DVO_ORACLE_ITERATIONS_EXCEEDED = 0,
  DVO_ORACLE_INCREMENT_TOO_SMALL = 1,
  DVO_ORACLE_LOGLIKELIHOOD_DECREASED = 2,
  DVO_ORACLE_TOO_FEW_CONSTRAINTS = 3
};

typedef struct {
  int32_t first_level, last_level;      /* dense_tracking_config.cpp:28-29 */
  int32_t max_iterations_per_level;
  int32_t use_initial_estimate;
  double precision;
  double mu;
  float intensity_derivative_threshold;
  float depth_derivative_threshold;
  int32_t mode;                         /* DVO_ORACLE_REF_SSE | DVO_ORACLE_MATH */
  int32_t reserved;
} oracle_config;

typedef struct {
  int32_t id, valid_constraints;
  double tdist_loglik;                  /* = -ll, dense_tracking.cpp:299 */
  double tdist_mean[2];
  double tdist_precision[4];            /* row-major 2x2 */
  double prior_loglik;
  double increment[6];                  /* x = (v, omega) */
  double information[36];               /* row-major 6x6, A incl. mu*I */
} oracle_iteration_stats;

typedef struct {
  int32_t id, max_valid_pixels, valid_pixels, termination, n_iterations, first_iteration_index;
} oracle_level_stats;

typedef struct {
  double transformation[16];            /* row-major 4x4, in: initial guess, out: estimate^-1 */
  double information[36];
  double loglik;
  int32_t n_levels, n_iterations_total;
} oracle_result;

typedef struct oracle_pyramid oracle_pyramid;

/* --- image model ------------------------------------------------------------------------- */
/* intensity: float 0..255, depth: float metres (NaN invalid); K = fx,fy,ox,oy of level 0 */
oracle_pyramid* oracle_pyramid_create(int width, int height, const float K[4],
                                      const float* intensity, const float* depth, int levels);
void oracle_pyramid_destroy(oracle_pyramid* p);
int oracle_pyramid_num_levels(const oracle_pyramid* p);
/* planes: 0=I 1=Z 2=Idx 3=Idy 4=Zdx 5=Zdy ; returns pointer to w*h floats (owned by pyramid) */
const float* oracle_pyramid_plane(oracle_pyramid* p, int level, int plane, int* w, int* h, float K[4]);
/* number of selected reference points at a level (point_selection.cpp:119-152) and, optionally, a
 * w*h uint8 mask of the selected pixels */
int oracle_pyramid_select(oracle_pyramid* p, int level, float ithr, float dthr, uint8_t* mask_or_null);

/* uint16 raw depth -> float metres, 0 -> NaN (surface_pyramid.cpp:65-105) */
void oracle_convert_raw_depth(const uint16_t* raw, float* out, int n, float scale);
/* BGR u8 -> grey float 0..255 with OpenCV's CV_BGR2GRAY fixed-point rounding (benchmark_slam.cpp:59-68) */
void oracle_bgr_to_grey(const uint8_t* bgr, float* out, int n);

/* --- the hot path ------------------------------------------------------------------------ */
/* full coarse-to-fine match (dense_tracking.cpp:131-376).  levels/iters are caller-allocated. */
int oracle_match(oracle_pyramid* ref, oracle_pyramid* cur, const oracle_config* cfg,
                 oracle_result* result,
                 oracle_level_stats* levels, int cap_levels,
                 oracle_iteration_stats* iters, int cap_iters);

/* independent matches on `nthreads` host threads, one match per thread at a time
 * (mirrors tbb::parallel_reduce over proposals, dvo_slam/src/keyframe_graph.cpp:576-593).
 * results[i].transformation is in/out. Returns wall seconds. */
double oracle_match_batch(int n, oracle_pyramid** refs, oracle_pyramid** curs, const oracle_config* cfg,
                          oracle_result* results, int nthreads);

typedef struct {
  int32_t n;                 /* valid constraints */
  int32_t n_selected;        /* reference points offered to the residual pass */
  float scale_cov[3];        /* C00 C01 C11 (already divided by n-3) */
  float precision[4];        /* P = C^-1 row-major */
  double neg_loglik;         /* -ll (float-rounded in REF_SSE like the reference) */
  double A[36];              /* J^T (w P) J, row-major, WITHOUT mu */
  double b[6];               /* -J^T (w P) r */
  double sum_w;              /* diagnostic: sum of weights */
} oracle_iteration_out;

/* one Gauss-Newton linearisation at a given float 3x4 estimate (row-major, maps reference points
 * into the current camera), i.e. passes 1-5 of dense_tracking.cpp:271-343 */
int oracle_level_iteration(oracle_pyramid* ref, oracle_pyramid* cur, int level, int mode,
                           float ithr, float dthr,
                           const float T34[12], const float P_prev[4], int first_iteration_on_level,
                           oracle_iteration_out* out,
                           float* residuals_or_null /* w*h*2, NaN where invalid */);

/* --- SE(3) / linear algebra helpers (exposed for unit tests) ------------------------------ */
void oracle_se3_exp(const double x[6], double T[16]);
void oracle_se3_log(const double T[16], double x[6]);
int oracle_solve6(const double A[36], const double b[6], double x[6]);

/* --- reduce-stage known-answer shape from dvo_core/src/sse_test.cpp:32-102 ----------------- */
/* J: n rows of 2x6 (row-major per point: J0[6], J1[6]); alpha 2x2 row-major symmetric.
 * mode REF_SSE: float sequential blocked accumulation (math_sse.cpp:82-178); MATH: float64. */
void oracle_rank_update_2x6(const float* J, int n, const float alpha[4], int mode, double A[36]);

/* The passes of one iteration on caller-supplied arrays, so that each can be pinned against the reference's own function
 * (oracle/_ref, tests/test_oracle_ref.py).  points: n x 12 floats (xyz1 + 8 channel values), accel: h x w x 8 floats,
 * T: row-major 3x4, P: row-major 2x2, C: {c00, c01, c11, c10}. */
int oracle_pass_residuals(int mode, int n, const float* points, const float* accel, int w, int h, const float K[4], const float T[12],
                          float* out_points, float* out_residuals);
void oracle_pass_weight_vectors(const float K[4], float reference_weight[8], float current_weight[8]);
void oracle_pass_weights(int mode, int n, const float* residuals, const float P[4], float* weights);
void oracle_pass_scale(int mode, int n, const float* residuals, const float* weights, float C[4] /* c00 c01 c11 c10 */);
double oracle_pass_loglik(int mode, int n, const float* residuals, const float P[4]);

const char* oracle_version(void);

#ifdef __cplusplus
}
#endif
#endif
