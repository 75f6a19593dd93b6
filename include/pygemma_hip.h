/*
 * pygemma_hip.h — C ABI of the MI355X-native pyGEMMA hot path (libpygemma_hip.so).
 *
 * The reference (rlangefe/pygemma) has NO C ABI: its operator boundary is the Python call
 * calculate((eigenVals, Y, W, X_block, grid)) -> list[dict]  (lmm/lmm.py:461-495) plus two
 * third-party calls, scipy.linalg.eigh(K) (lmm/lmm.py:152,197) and U.T @ X (lmm/lmm.py:244-246).
 * This header DEFINES the boundary a maintainer would bind with ctypes (INTEGRATION.md shows the
 * stub).  Plain C: pointers + sizes, no C++/torch types.
 *
 * Conventions
 *  - every function returns 0 on success, a negative PG_E* code on failure; never throws, never
 *    calls exit(); pg_last_error() returns a thread-local message for the last failure.
 *  - "_dev" entry points take DEVICE pointers (hipMalloc'd, or torch tensors' data_ptr()) and
 *    enqueue on the context's stream; they return after enqueueing (call pg_ctx_sync).
 *    Entry points without "_dev" take HOST pointers, copy in/out and return when done.
 *  - the caller owns every buffer it passes; the library owns only what hangs off pg_ctx.
 *  - layouts: "SNP-major" Xr[g*ldx + i] (row g = rotated genotype vector of SNP g, ldx >= n);
 *    "reference layout" X[i*p + g] (NumPy C-order (n,p), lmm/lmm.py:121-122).
 */
#ifndef PYGEMMA_HIP_H
#define PYGEMMA_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PG_OK 0
#define PG_EINVAL (-22)     /* bad argument (shape, NULL, unsupported c)               */
#define PG_ENOMEM (-12)     /* device allocation failed                                */
#define PG_EHIP (-5)        /* a HIP runtime call or kernel launch failed               */
#define PG_ENOTSUP (-95)    /* configuration not supported by this build                */
#define PG_ENODEV (-19)     /* no usable GPU                                            */

#define PG_MAX_COVARIATES 30 /* c supported by the register-resident Gram kernels (reference benchmark: up to 26) */

typedef struct pg_ctx pg_ctx; /* one per (process, GPU): device id, stream, scratch      */

const char *pg_last_error(void);
const char *pg_version(void);
int pg_device_count(void);                      /* GPUs visible to HIP (0 if none)       */
int pg_ctx_create(int device, pg_ctx **out);    /* binds a device, creates a stream       */
int pg_ctx_create_on_stream(int device, void *hip_stream, pg_ctx **out); /* caller's stream */
void pg_ctx_destroy(pg_ctx *ctx);
int pg_ctx_sync(pg_ctx *ctx);
int pg_ctx_device(const pg_ctx *ctx);

/* device memory helpers for callers without their own allocator (Python/ctypes hosts) */
int pg_malloc(pg_ctx *ctx, size_t bytes, void **dptr);
int pg_free(pg_ctx *ctx, void *dptr);
int pg_mem_info(pg_ctx *ctx, size_t *free_bytes, size_t *total_bytes);   /* of ctx's device (how much of X may be prefetched) */
int pg_memcpy_h2d(pg_ctx *ctx, void *dst, const void *src, size_t bytes);
int pg_memcpy_d2h(pg_ctx *ctx, void *dst, const void *src, size_t bytes);
int pg_memset(pg_ctx *ctx, void *dst, int value, size_t bytes);
/* strided host -> device copy: `height` rows of `width` bytes (a column window of a row-major host matrix) */
int pg_memcpy2d_h2d(pg_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height);

/* ---- S1: pinned host memory + asynchronous copies: streaming SNP batches and precomputed eigenvectors from the host
 * (BASELINE configs 4-5).  The reference's eigen=False caller reads raw float32 .bin files and hands the whole pre-rotated X
 * over (experiments/large_gwas/run_pygemma.py:33-65); here X is read straight out of pinned memory by the DMA engines, batch
 * by batch, while the previous batch computes.
 *   pg_host_alloc/free        : hipHostMalloc'd (portable) buffer — e.g. the array a caller np.fromfile()s its .bin into
 *   pg_host_register/unregister: pin a caller-owned range in place
 *   pg_memcpy*_async          : enqueue on the context's stream and return (host side should be pinned; pageable memory
 *                               makes the runtime stage the copy and blocks)
 *   pg_stage_rows             : host-side gather of a column window into a (pinned) staging buffer with nthreads threads —
 *                               the pageable -> pinned leg for callers whose X is an ordinary NumPy array */
int pg_host_alloc(pg_ctx *ctx, size_t bytes, void **hptr);
int pg_host_free(pg_ctx *ctx, void *hptr);
int pg_host_register(pg_ctx *ctx, void *hptr, size_t bytes);
int pg_host_unregister(pg_ctx *ctx, void *hptr);
int pg_memcpy_h2d_async(pg_ctx *ctx, void *dst, const void *src, size_t bytes);
int pg_memcpy_d2h_async(pg_ctx *ctx, void *dst, const void *src, size_t bytes);
int pg_memcpy_d2d_async(pg_ctx *ctx, void *dst, const void *src, size_t bytes);
int pg_memcpy2d_h2d_async(pg_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height);
int pg_stage_rows(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, int nthreads);

/* HIP events on the context's stream (kernel timing for bench.py's roofline leg) */
int pg_event_create(pg_ctx *ctx, void **event);
int pg_event_destroy(pg_ctx *ctx, void *event);
int pg_event_record(pg_ctx *ctx, void *event);
int pg_event_elapsed_ms(pg_ctx *ctx, void *start, void *stop, float *ms); /* synchronises on `stop` */
int pg_event_sync(pg_ctx *ctx, void *event);            /* host waits for the event                         */
int pg_stream_wait_event(pg_ctx *ctx, void *event);     /* the context's stream waits for it (no host wait) */

/* ---- H3 / SURVEY 8e: the one exchange step of the path, RCCL over xGMI called directly (librccl is dlopen'ed on first
 * use; no PyTorch).  Replaces Pool(nproc).imap + the ordered concatenation of the per-block lists (lmm/lmm.py:378-403):
 * a SNP block lives on one GPU; U/d/rotated y,W travel once (broadcast from the GPU that ran the eigensolver), the 32-byte
 * result rows once at the end (all-gather in rank order = SNP order, blocks padded to ceil(p/G) rows like SampleIter's).
 *   pg_comm_unique_id + pg_comm_init_rank : one process per GPU; rank 0 makes the 128-byte id and passes it to the other
 *                                           ranks over any host channel (pygemma_amd/dist.py: file or TCP rendezvous)
 *   pg_comm_init_all                      : one process, ndev GPUs, one host thread per GPU issuing the collectives
 * Collectives are enqueued on the stream of the context the communicator was made with. */
typedef struct pg_comm pg_comm;
#define PG_COMM_ID_BYTES 128
int pg_comm_unique_id(void *id128);
int pg_comm_init_rank(pg_ctx *ctx, int nranks, int rank, const void *id128, pg_comm **out);
int pg_comm_init_all(int ndev, pg_ctx *const *ctxs, pg_comm **out /* [ndev] */);
int pg_comm_destroy(pg_comm *comm);
int pg_comm_size(const pg_comm *comm);
int pg_comm_rank(const pg_comm *comm);
int pg_comm_broadcast_dev(pg_comm *comm, void *buf, size_t bytes, int root);                      /* in place */
int pg_comm_allgather_dev(pg_comm *comm, const void *send, void *recv, size_t bytes_per_rank);
int pg_comm_allreduce_f64_dev(pg_comm *comm, double *buf, size_t count, int op_max);              /* in place; sum or max */
int pg_comm_barrier(pg_comm *comm);                                                               /* + stream sync */
int pg_comm_group_start(void);
int pg_comm_group_end(void);

/* ---- H3-H10: the per-SNP operator -------------------------------------------------------
 * Replaces calculate() (lmm/lmm.py:461-495) and everything under it: calc_lambda_restricted
 * (pygemma_model.pyx:64-194, grid :99-132 / decade-scan+brentq+newton :135-194), precompute_mat
 * (:880-1053), newton (:1349-1416), the *_overload scalars (:1656-1698, :1813-1830),
 * calc_beta_vg_ve_restricted_overload (:1514-1537) and scipy.stats.f.sf (lmm/lmm.py:482).
 *
 * Inputs are in the eigenbasis (the reference's eigen=False boundary, lmm/lmm.py:164-167):
 *   d   (n)      eigenvalues, already clamped >= 0
 *   Wr  (n x c)  rotated covariates, row-major (NumPy C-order)
 *   yr  (n)      rotated phenotype
 *   Xr  (p rows) rotated genotypes, SNP-major, row stride ldx >= n
 * Outputs (length p, caller-allocated, SNP order preserved): beta, se_beta, tau, lambda as
 * float32; F_wald, p_wald as float64 (lmm/lmm.py:476-483; lambda is the f32 value the reference
 * widens to a Python float).  pval may be NULL.  grid != 0 selects calc_lambda_restricted's grid
 * branch.  stats (may be NULL) receives [fast evaluations, full (Newton) evaluations] summed over SNPs.
 */
int pg_assoc_dev(pg_ctx *ctx, int64_t n, int c, int64_t p, const float *d, const float *Wr, const float *yr,
                 const float *Xr, int64_t ldx, int grid, float *beta, float *se, float *tau, float *lambda,
                 double *F, double *pval, unsigned long long *stats_dev);

/* Work-per-SNP trace (tail analysis of the data-dependent Brent/Newton path, pygemma_model.pyx:1349-1416): the following
 * pg_assoc_dev calls on this context write, per SNP, fast evaluations | full (Newton) evaluations << 16 into trace_dev
 * (device, >= p entries, caller-owned); NULL switches the trace off. */
int pg_assoc_set_eval_trace(pg_ctx *ctx, unsigned *trace_dev);
/* Optional: what the FIRST pg_assoc_dev / pg_assoc_lrt_dev call of a context does on the host before it can enqueue its kernels (the
 * float32-sum plan of numpy's add.reduce for length n: built on the host, uploaded after a stream drain; scratch allocations for c
 * covariates) — ahead of time.  A streaming caller (lmm/lmm.py:461-495's per-SNP loop cut into batches) calls it while its first batch is
 * still on the host link.  No reference counterpart: the reference has no device. */
int pg_assoc_warm(pg_ctx *ctx, int64_t n, int c);

/* ---- N2: the same operator plus the likelihood-ratio test the reference sketches and leaves commented out
 * ("Fix these calculations later", lmm/lmm.py:137-141, 277-300), built from its own ML functions: lambda_alt =
 * calc_lambda(eigenVals, Y, [W, x]) (lmm/lmm.py:22-84: decade scan of dlogL/dlambda, brentq(rtol=0.1) + scipy newton),
 * l_alt = likelihood_lambda(lambda_alt, ...) (pygemma_model.pyx:1542-1562; derivatives :1567-1603), l_null the same for the
 * covariates alone, D_lrt = 2 (l_alt - l_null) on float32 scalars, p_lrt = chi2(1).sf(D_lrt).  The first six outputs are
 * bit-identical to pg_assoc_dev's.  l_alt, l_null (the same value in every row), D_lrt: float32 values widened to float64
 * (like the frame's lambda column); p_lrt float64.  The reference evaluates the quadratic forms of its ML functions with
 * float32 NumPy helpers (pyx:2045-2180); here they come from the same float64 sweeps as the REML path, so agreement with the
 * reference is at its float32 noise (l within 2 ulp, D within 2.5e-4: tests/golden/lrt_panels.npz), not bit-for-bit. */
int pg_assoc_lrt_dev(pg_ctx *ctx, int64_t n, int c, int64_t p, const float *d, const float *Wr, const float *yr,
                     const float *Xr, int64_t ldx, int grid, float *beta, float *se, float *tau, float *lambda,
                     double *F, double *pval, double *l_alt, double *l_null, double *D_lrt, double *p_lrt);

/* host-pointer convenience: X in the REFERENCE layout (n x p row-major, already rotated), as
 * calculate() receives it; transposed to SNP-major on the device. */
int pg_assoc(pg_ctx *ctx, int64_t n, int c, int64_t p, const float *d, const float *Wr, const float *yr,
             const float *X_n_by_p, int grid, float *beta, float *se, float *tau, float *lambda, double *F,
             double *pval, unsigned long long *stats2);

/* pg_assoc over several GPUs of the node: contiguous blocks of ceil(p/ngpu) SNP columns like the reference's
 * SampleIter (lmm/lmm.py:427-434) — one host thread and one context per GPU; the 32-byte result rows of all blocks are
 * all-gathered over RCCL (pg_comm_*) and leave the node's GPU 0 in SNP order with one copy per column. */
int pg_assoc_multi(int ngpu, int64_t n, int c, int64_t p, const float *d, const float *Wr, const float *yr,
                   const float *X_n_by_p, int grid, float *beta, float *se, float *tau, float *lambda, double *F, double *pval);

/* scipy.stats.f.sf(F, 1, dfd) (lmm/lmm.py:482) for a device vector */
int pg_fdist_sf_dev(pg_ctx *ctx, int64_t count, const double *F, double dfd, double *pval);

/* (n x p row-major, row stride ldX >= p) -> SNP-major (p x ldx), pad columns [n, ldx) zero-filled */
int pg_transpose_dev(pg_ctx *ctx, int64_t n, int64_t p, const float *X_n_by_p, int64_t ldX, float *Xr, int64_t ldx);

/* ---- H2: rotation  X <- U' X  (lmm/lmm.py:243-246, OpenBLAS sgemm in the reference) ---------
 * U (n x n, row stride ldU) row-major with eigenvector j in COLUMN j (scipy.linalg.eigh's convention); X in
 * the reference layout (n x p, row stride ldX >= p: a column window of a wider matrix is fine); output
 * SNP-major Xr (p x ldx): Xr[g*ldx + k] = sum_i U[i*ldU + k] * X[i*ldX + g], pad columns [n, ldx) zeroed.
 * fp32 MFMA (v_mfma_f32_32x32x2_f32), fp32 accumulate in sample order i = 0..n-1 — the reference's precision. */
int pg_rotate_dev(pg_ctx *ctx, int64_t n, int64_t p, const float *U, int64_t ldU, const float *X_n_by_p, int64_t ldX,
                  float *Xr, int64_t ldx);

/* ---- N4: rotation fast path for GENOTYPE columns (each column of the block takes <= 3 equally spaced
 * values: hard calls 0/1/2, raw or centred/standardised).  U'x = v0 (U'1) + dx (U'code): the codes are exact in int8,
 * U is held once as a 24-bit fixed point per eigenvector in three int8 digit planes (|error| <= 2^-24 * 2 max_i |U[i][k]|:
 * float32's own rounding for entries within a factor two of the column's largest), products and sums are EXACT on the int8
 * MFMA pipe, and the three partial sums meet in fp64 with one rounding to float32 (error below pg_rotate_dev's / the
 * reference's sgemm's fp32 accumulation on delocalised eigenvectors, within 3x of it on the adversarial ones of
 * tests/test_gpu_rotate.py; 1/7 of its time).  PG_GENO_I8=0 selects the r2-r3 kernel instead: U split into two fp16 planes by
 * round-to-nearest (residual <= 2^-23 |U|), fp32 accumulation on the fp16 MFMA pipe.  A column may also hold ONE other
 * value anywhere (missing calls imputed with the column mean, experiments/benchmarks/benchmarks.py:243-244): such blocks
 * take a second, accumulating pass on the 0/1 indicator plane.
 * Finite blocks that are not genotype-valued (imputed dosages, any float X) take the same GEMM with X itself split into two
 * fp16 planes (per-column power-of-two scale, residual <= 2^-23 |x|): two passes, still 3x faster than pg_rotate_dev.
 *   pg_geno_prep_dev   : once per U -> Uprep (pg_geno_prep_bytes(n) bytes, device)
 *   pg_rotate_geno_dev : per SNP block; Xr written (same layout as pg_rotate_dev) and *is_geno = 1 (genotype-valued block)
 *                        or 2 (general finite block, split path); *is_geno = 0 and Xr untouched when the block holds a NaN
 *                        or Inf (call pg_rotate_dev: its propagation is the reference's).  Synchronises the stream. */
size_t pg_geno_prep_bytes(int64_t n);
size_t pg_geno_work_bytes(int64_t n, int64_t p);
int pg_geno_prep_dev(pg_ctx *ctx, int64_t n, const float *U, int64_t ldU, void *Uprep);
int pg_rotate_geno_dev(pg_ctx *ctx, int64_t n, int64_t p, const void *Uprep, const float *X_n_by_p, int64_t ldX, float *Xr,
                       int64_t ldx, void *work, int *is_geno);
/* pg_rotate_geno_dev + the caller's fallback to pg_rotate_dev in ONE enqueue, with the path chosen on the device: every candidate
 * kernel is launched predicated on the flags of the detect pass, so the stream is never synchronised (the flag read-back of
 * pg_rotate_geno_dev idles the GPU for ~0.7 ms per 16 384-SNP block at n = 10 000).  float32 X only.  U (row stride ldU) is the
 * operand of the fp32-MFMA fallback; path_dev (device int, may be NULL) receives 1 / 2 / 0 like *is_geno. */
int pg_rotate_auto_dev(pg_ctx *ctx, int64_t n, int64_t p, const float *U, int64_t ldU, const void *Uprep, const float *X_n_by_p, int64_t ldX,
                       float *Xr, int64_t ldx, void *work, int *path_dev);
int pg_rotate_auto_i8_dev(pg_ctx *ctx, int64_t n, int64_t p, const void *Uprep, const void *X8_n_by_p, int is_unsigned, int64_t ldX,
                          float *Xr, int64_t ldx, void *work, int *path_dev);   /* int8/uint8 X (always finite): genotype or split path */
/* The same for X stored as 8-bit integers (int8 / uint8 genotype matrices; the reference casts any dtype to float32,
 * lmm/lmm.py:121-122, so the values are identical): 4x fewer bytes to upload and to scan.  pg_cast_i8_f32_dev makes the
 * float32 image a block needs when it does not qualify (then pg_rotate_dev as usual). */
int pg_rotate_geno_i8_dev(pg_ctx *ctx, int64_t n, int64_t p, const void *Uprep, const void *X8_n_by_p, int is_unsigned, int64_t ldX,
                          float *Xr, int64_t ldx, void *work, int *is_geno);
int pg_cast_i8_f32_dev(pg_ctx *ctx, int64_t n, int64_t p, const void *X8_n_by_p, int is_unsigned, int64_t ldX, float *Xf, int64_t ldXf);
/* ... and for float64 X (numpy's default): the reference's X.astype(np.float32) is a per-element round-to-nearest, which the
 * kernels apply as they read the block — no host-side float32 copy of the matrix. */
int pg_rotate_geno_f64_dev(pg_ctx *ctx, int64_t n, int64_t p, const void *Uprep, const double *X_n_by_p, int64_t ldX, float *Xr,
                           int64_t ldx, void *work, int *is_geno);
int pg_cast_f64_f32_dev(pg_ctx *ctx, int64_t n, int64_t p, const double *X_n_by_p, int64_t ldX, float *Xf, int64_t ldXf);
/* The same rotation straight from a PLINK .bed block (the format the reference's callers read with pysnptools.Bed,
 * experiments/benchmarks/benchmarks.py:233-239): bed = device copy of p SNP records of ldb >= ceil(n/4) bytes (SNP-major,
 * 2 bits per sample: 00 hom A1, 01 missing, 10 het, 11 hom A2).  Dosage = copies of A2 (count_a1 = 0, pysnptools
 * count_A1=False) or of A1; missing calls take the mean of the called genotypes of their SNP (SimpleImputer 'mean',
 * benchmarks.py:243-244).  16x fewer input bytes than float32 X.  work: pg_geno_work_bytes(n, p). */
int pg_rotate_bed_dev(pg_ctx *ctx, int64_t n, int64_t p, const void *Uprep, const unsigned char *bed, int64_t ldb, int count_a1,
                      float *Xr, int64_t ldx, void *work);

/* ---- N3: relatedness matrix from standardised genotypes, K = G G' / p_k
 * (experiments/animal_gwas/run_gwas.py:45-55, tests/test_pygemma.py:184-192).  Gt is the SNP-major (p_k x ldg)
 * image of the standardised G (n x p_k), e.g. from pg_transpose_dev; K (n x n, row-major) float32, both triangles written
 * (computed as a syrk: tiles on or below the diagonal only). */
int pg_kinship_dev(pg_ctx *ctx, int64_t n, int64_t p_k, const float *Gt, int64_t ldg, float *K);
/* The same straight from the (n x p) genotype matrix G (row-major, row stride ldG), as calculate_genetic_relatedness_matrix
 * builds it (run_gwas.py:46-56): standardize != 0 centres every column and divides by its population standard deviation
 * (fp64 statistics, sd == 0 -> 1) on the device; then a lower-triangle syrk on the fp32 MFMA pipe (the upper triangle is the
 * mirror, bit-symmetric).  K can go straight into pg_syevd_dev. */
int pg_kinship_geno_dev(pg_ctx *ctx, int64_t n, int64_t p, const float *G, int64_t ldG, int standardize, float *K);

/* ---- lmm/lmm.py:124-125: K <- Z K Z' for the optional design matrix Z (n x q) of the random effect, K (q x q).  Z and K may each be
 * float32 or float64 (z_is_f64 / k_is_f64), row-major with row strides ldz / ldk, on the device; out = float32 (n x n, row stride ldo),
 * what lmm.py:127-128 hands to the eigensolver.  Two fp64-MFMA products and one rounding. */
int pg_zkzt_dev(pg_ctx *ctx, int64_t n, int64_t q, const void *Z, int z_is_f64, int64_t ldz, const void *K, int k_is_f64, int64_t ldk,
                float *out, int64_t ldo);

/* ---- H1: eigendecomposition of K (lmm/lmm.py:151-162 / :196-207, scipy.linalg.eigh = LAPACK ssyevr)
 * Reads the LOWER triangle of row-major K (n x n, float32, device).  Computes in fp64; delivers ascending
 * eigenvalues clamped at 0 (lmm/lmm.py:157) as float32, and U (column j = eigenvector j) as float32
 * (and optionally fp64 for the invariant checks: U64/evals64 may be NULL).
 * Two reductions to tridiagonal form behind this one entry point: from n = 768 on, dense -> band (fp64 MFMA GEMMs) -> tridiagonal
 * (persistent bulge-chasing kernel) with two blocked back-transformations (csrc/sb2.hip); below that, and for a K whose panels the
 * band reduction cannot factor (rank-deficient K: decided on the device), the one-stage Householder reduction (csrc/syevd.hip).
 * Environment, for tests and A/B timing only: PG_SYEVD_STAGES=1|2 forces a path, PG_SYEVD_TIMING=1 prints phase times.
 * Work space ~ 10 n^2 doubles on the two-stage path (200 GB at n = 50 000; when the device cannot give it the one-stage path is taken),
 * ~ 8 n^2 on the one-stage path. */
int pg_syevd_dev(pg_ctx *ctx, int64_t n, const float *K, float *evals, float *U, double *evals64, double *U64);

/* ---- Inspection surface: the model-level functions the reference's tests call directly
 * (tests/test_pygemma.py:256-294).  One wavefront each; for fixture-level parity, not for throughput.
 * All pointers are device pointers.  ctot = columns of Wx = the reference's np.c_[W, x] (SNP last), m = ctot + 1.
 *
 * pg_precompute_mat_dev : precompute_mat(lam, eigenVals, W, Y, full) (pygemma_model.pyx:880-1053).
 *     P3/Q3/R3 : m*m*m float32 each, indexed [row][level][col] like wjt_Pi_wk / wjt_Pi_Pi_wk / wjt_Pi_Pi_Pi_wk;
 *                entries the reference leaves undefined (np.empty) are NaN; R3 all NaN unless full.
 *     vecs     : [5][m] = yt_Pi_y, yt_Pi_Pi_y, yt_Pi_Pi_Pi_y, tr_Pi, tr_Pi_Pi per level.
 *     scal     : 8 floats = logdet_Wt_H_inv_W, logdet_H, then d1, d2 (NaN unless full), logL at the last level, sum h, sum h^2
 *                (NaN unless full): the un-projected traces the ML functions use.
 * pg_newton_dev         : newton(lam, eigenVals, Y, W, precompute=True, lambda_min, lambda_max) (pyx:1349-1416) -> *root.
 * pg_reml_scalars_dev   : args8 = {lam, yPy, yPPy, yPPPy, trP, trPP, logdet_H, logdet_Wt_H_inv_W} ->
 *     out3 = {likelihood_restricted_lambda_overload (pyx:1813), likelihood_derivative1_..._overload (pyx:1656),
 *             likelihood_derivative2_..._overload (pyx:1675)} with n and c = ctot as the reference passes them. */
int pg_precompute_mat_dev(pg_ctx *ctx, int64_t n, int ctot, float lam, const float *d, const float *Wx, const float *y,
                          int full, float *P3, float *Q3, float *R3, float *vecs, float *scal);
int pg_newton_dev(pg_ctx *ctx, int64_t n, int ctot, float lam, float lam_min, float lam_max, const float *d,
                  const float *Wx, const float *y, float *root);
int pg_reml_scalars_dev(pg_ctx *ctx, int64_t n, int ctot, const float *args8, float *out3);
/* pg_ml_scalars_dev     : args7 = {lam, yPy, yPPy, yPPPy, sum h, sum h^2, logdet_H} -> out3 = {likelihood_lambda (pyx:1542),
 *     likelihood_derivative1_lambda (pyx:1567), likelihood_derivative2_lambda (pyx:1586)}: the ML scalars of the LRT (N2). */
int pg_ml_scalars_dev(pg_ctx *ctx, int64_t n, const float *args7, float *out3);

/* ---- Test hooks of the eigensolver's stages (tests/test_gpu_syevd.py, tools/bench_dgemm.py): not part of the drop-in
 * surface, exported so that each stage can be checked against host LAPACK on its own.
 * pgx_dgemm_dev : C = alpha op(A) B + beta C in fp64 on v_mfma_f64_16x16x4_f64 (row-major; transA: A is K x M)
 * pgx_sytrd_dev : Householder tridiagonalisation of the lower triangle of float32 K -> d, e, tau, reflectors (device)
 * pgx_stedc_dev : divide & conquer on a tridiagonal given on the host -> eigenvalues (host), eigenvectors Z (device) */
int pgx_dgemm_dev(pg_ctx *ctx, int transA, int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t lda,
                  const double *B, int64_t ldb, double beta, double *C, int64_t ldc);
/* pgx_dgemm_ex_dev: every mode of that GEMM: flags bit 0 transA, bit 1 transB (B stored N x K), bit 2 lower triangle of C only,
 * bit 3 A symmetric (lower triangle + full diagonal tiles valid), bit 4 no split-K; kxorB: B's k index XOR-ed (multiple of 8) */
int pgx_dgemm_ex_dev(pg_ctx *ctx, int flags, int kxorB, int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t lda,
                     const double *B, int64_t ldb, double beta, double *C, int64_t ldc);
/* pgx_ring_stamps: in-kernel time stamps one workgroup of the ring GEMM left when PG_DGEMM_TUNE has bit 3 set (diagnostics) */
int pgx_ring_stamps(long long *out64);
int pgx_sytrd_dev(pg_ctx *ctx, int64_t n, const float *K, double *d, double *e, double *tau, double *Vall);
int pgx_stedc_dev(pg_ctx *ctx, int64_t n, const double *d_host, const double *e_host, double *evals_host, double *Z_dev);
/* The two-stage tridiagonalisation of pg_syevd_dev (csrc/sb2.hip), one stage at a time (all matrices n x n fp64 row-major on the device):
 * pgx_sb2_stage1_dev : lower triangle of float32 K -> Aband (the band |i - j| <= 64 of the result is the band matrix Q1' K Q1);
 *                      Z, if given, is replaced by Q1 Z.  flags (host, 4 ints): [0] a panel needed the one-stage fallback.
 * pgx_sb2_stage2_dev : band |i - j| <= 64 of Aband (lower part read) -> tridiagonal d (n), e (n - 1) on the device by bulge chasing;
 *                      Z, if given, is replaced by Q2 Z.  flags: [1] a wait inside the bulge-chasing kernel expired. */
int pgx_sb2_stage1_dev(pg_ctx *ctx, int64_t n, const float *K, double *Aband, double *Z, int *flags);
int pgx_sb2_stage2_dev(pg_ctx *ctx, int64_t n, const double *Aband, double *d, double *e, double *Z, int *flags);
/* debugging aid: host-mapped memory (pg_host_alloc, >= 4 ints) in which the bulge-chasing kernel records (sweep, step, phase); NULL = off */
int pgx_sb2_set_debug(void *host_mapped);

#ifdef __cplusplus
}
#endif
#endif /* PYGEMMA_HIP_H */
