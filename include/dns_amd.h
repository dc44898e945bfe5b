/*
 * dns_amd.h -- C-ABI of the MI355X (gfx950) saddle-point time-stepping path.
 *
 * Drop-in boundary for dolfin_navier_scipy's linear-algebra layer.  The
 * reference reaches its solver through the Python module
 * `sadptprj_riclyap_adi.lin_alg_utils` (un-vendored; call sites listed below)
 * and through `scipy.sparse.linalg.factorized`; the Python wrapper
 * `dolfin_navier_scipy_amd/lin_alg_utils.py` binds the entry points declared
 * here with `ctypes` and presents that same Python surface.
 *
 * Conventions
 *  - plain pointers and sizes only; all pointers are HOST pointers unless the
 *    name ends in `_dev`; buffers are borrowed for the duration of the call;
 *  - matrices are CSR with int32 indices and float64 values (what
 *    `dolfin_to_sparrays.mat_dolfin2sparse` produces, reference dts:67-81);
 *  - every function returns a status code (0 = ok); `dns_last_error()` gives
 *    the message of the last failure on the calling thread;
 *  - one host thread per handle, one HIP stream per handle.
 *
 * Saddle-point contract:
 *      [ F   JT ] [ v  ]   [ rhs_v ]
 *      [ J   0  ] [ p~ ] = [ rhs_p ]      F = M + theta*dt*(A [+ N(v_lin)])
 */
#ifndef DNS_AMD_H
#define DNS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DNS_OK               0
#define DNS_NOT_CONVERGED    1   /* maxiter reached, best iterate returned   */
#define DNS_BREAKDOWN        2   /* Krylov / factorisation breakdown         */
#define DNS_ERR_HIP          3   /* a HIP runtime call failed                */
#define DNS_ERR_BAD_ARGUMENT 4
#define DNS_ERR_NOT_READY    5   /* e.g. solve before the preconditioner     */
#define DNS_ERR_COMM         6   /* RCCL failure                             */
#define DNS_ERR_HOST         7   /* host-side failure inside the library (out
                                    of memory, a thread that cannot start):
                                    no C++ exception crosses this boundary   */

#define DNS_METHOD_GMRES     0
#define DNS_METHOD_BICGSTAB  1

#define DNS_SCHUR_DENSE      0   /* explicit inverse of J Fh^-1 JT (NP small) */
#define DNS_SCHUR_JACOBI     1   /* diag(J D^-1 JT)^-1  (always available)   */
#define DNS_SCHUR_MG         2   /* one multigrid V-cycle on the sparse Schur
                                    complement (dns_saddle_set_schur_mg)     */

#define DNS_SPMV_VECTOR      0   /* sub-wave per row, shuffle reduction      */
#define DNS_SPMV_STREAM      1   /* row blocks streamed through LDS          */
#define DNS_SPMV_STREAM16    2   /* the same, 16-bit column offsets per block */

typedef struct dns_saddle dns_saddle;   /* opaque: one saddle-point system  */
typedef struct dns_imex dns_imex;       /* opaque: device-resident stepper  */

typedef struct dns_csr {         /* host-side CSR view (borrowed)            */
    int32_t nrows, ncols;
    int64_t nnz;
    const int32_t *rowptr;       /* nrows+1 */
    const int32_t *colidx;       /* nnz     */
    const double  *vals;         /* nnz     */
} dns_csr;

#define DNS_FHAT_CHEB        0   /* Chebyshev recurrence: degree-1 SpMVs/apply */
#define DNS_FHAT_EXPLICIT    1   /* same polynomial as ONE explicit CSR matrix */
#define DNS_FHAT_AUTO        2   /* explicit while the system is launch-bound  */

/* block structure of the right preconditioner (both use the same Fh^-1 and the
 * same Sh = J Fh^-1 JT):
 *   TRIANGULAR  P = [[Fh, JT], [0, -Sh]]            one Fh^-1 apply
 *   FULL        P = [[Fh, JT], [J, 0]] exactly (block LDU):
 *               zp = -Sh^-1 (rp - J Fh^-1 rv),  zv = Fh^-1 (rv - JT zp)
 *               one more sparse product (J Fh^-1, formed explicitly) per
 *               apply; needs the explicit Fh^-1 and one GPU.  With a residual
 *               whose pressure part vanishes (warm-started time steps) the
 *               triangular form spends two Krylov steps before it starts to
 *               converge; the full form gains ~2 orders in its first step.  */
#define DNS_FACT_TRIANGULAR  0
#define DNS_FACT_FULL        1

typedef struct dns_precond_opts {
    int32_t cheb_degree;         /* terms of the Jacobi-Chebyshev F^-1 (>=1) */
    int32_t schur;               /* DNS_SCHUR_*                              */
    int32_t fhat;                /* DNS_FHAT_*                               */
    int32_t fp32_store;          /* 1: keep the explicit preconditioner
                                    matrices (Gc values, Schur inverse) in
                                    fp32 -- halves their bytes; the Krylov
                                    iteration itself stays fp64             */
    double  eig_lo_safety;       /* multiply the estimated lambda_min (0.9)  */
    double  eig_hi_safety;       /* multiply the estimated lambda_max (1.05) */
    double  eig_lo, eig_hi;      /* >0: use these bounds, skip the estimate  */
    double  drop_tol;            /* explicit Gc: drop |g_ij| < tol*max_j|g_ij| */
    int32_t factorization;       /* DNS_FACT_*                               */
    int32_t pad;
} dns_precond_opts;

typedef struct dns_solve_opts {
    int32_t method;              /* DNS_METHOD_*                             */
    int32_t restart;             /* GMRES cycle length (<= 64)               */
    int32_t maxiter;             /* total inner iterations                   */
    int32_t reorth;              /* 0: classical Gram-Schmidt once, 1: twice,
                                    2: once, folded into the next step's head
                                       kernel (norm by Pythagoras; one GPU) */
    double  rtol;                /* stop at ||r|| <= max(rtol*||b||, atol)   */
    double  atol;
    int32_t check_every;         /* host polls the device flag every k its   */
    int32_t use_graph;           /* replay hipGraph chunks instead of eager  */
} dns_solve_opts;

typedef struct dns_solve_stats {
    int32_t iters;               /* Krylov iterations performed              */
    int32_t status;              /* DNS_OK / DNS_NOT_CONVERGED / ...         */
    int32_t spmv_count;          /* K applies + F applies (preconditioner)   */
    int32_t restarts;
    double  bnorm;               /* ||[rhs_v; rhs_p]||_2                     */
    double  est_relres;          /* recurrence residual / ||b||              */
    double  true_relres;         /* ||b - K x|| / ||b|| recomputed at the end*/
    double  device_seconds;      /* HIP-event time of the solve on its stream*/
} dns_solve_stats;

/* ---- library ---------------------------------------------------------- */
int         dns_version(void);
const char *dns_status_string(int status);
const char *dns_last_error(void);
int         dns_device_count(int *count);
int         dns_device_name(int device, char *buf, size_t buflen);
/* block until all work queued on `device` has finished */
int         dns_device_synchronize(int device);

/* ---- saddle-point systems ---------------------------------------------
 * Replaces, per call site of `lau.solve_sadpnt_smw(amat=, jmat=, jmatT=,
 * rhsv=, rhsp=, krylov=, krpslvprms=, ...)`:
 *   stokes_navier_utils.py:401,458,497 (steady), :894,904 (initial Stokes),
 *   :1505-1512 (per time step), :1622,1629 (get_pfromv);
 *   time_int_utils.py:402,408,466 (Heun start), :605 (return_alu);
 * and `scipy.sparse.linalg.factorized(K)` + `coeffmatlu(rhs)` at
 *   time_int_utils.py:89-91,134 and :304-306,348.
 * `jt` may be NULL (the transpose of `j` is formed on the host, as
 * `jmatT=None -> jmat.T` at time_int_utils.py:605).
 */
int dns_saddle_create(int device, const dns_csr *f, const dns_csr *j,
                      const dns_csr *jt, dns_saddle **out);
void dns_saddle_destroy(dns_saddle *h);

/* same sparsity pattern, new values of F (Newton/Picard re-linearisation:
 * `solvmat = M + 0.5*dt*(A + N(v))`, stokes_navier_utils.py:1034,1484-1491) */
int dns_saddle_update_values(dns_saddle *h, const double *f_vals);

/* build / rebuild the block preconditioner  P = [[Fh, JT], [0, -Sh]] */
int dns_saddle_setup_precond(dns_saddle *h, const dns_precond_opts *opts);

/* out_vp = [v; p~] (NV+NP); x0 may be NULL (`krpslvprms['x0']`,
 * stokes_navier_utils.py:1493-1503); rhs_p may be NULL (zero,
 * stokes_navier_utils.py:1629-1632) */
int dns_saddle_solve(dns_saddle *h, const double *rhs_v, const double *rhs_p,
                     const double *x0, double *out_vp,
                     const dns_solve_opts *opts, dns_solve_stats *stats);

/* `ncols` right-hand sides in one call -- `lau.solve_sadpnt_smw` takes
 * `(NV, k)` blocks (`umat` / `vmat` feedback, stokes_navier_utils.py:1036-1042,
 * 1512; `apply_massinv`, tests/time_dep_nse_bigchannel.py:33): the blocks cross
 * the PCIe once, the solves run back to back on the resident system.
 * Column c of `rhs_v` at rhs_v + c NV, of `rhs_p` (may be NULL) at rhs_p + c NP,
 * of `out_vp` at out_vp + c (NV + NP); `x0`: `x0_cols` start vectors (0 none,
 * 1 shared, `ncols` one each); `stats` (may be NULL) holds `ncols` records.
 * A column that does not converge is reported in its record, like the single
 * solve; the residual histories per column: dns_saddle_residual_history_col */
int dns_saddle_solve_multi(dns_saddle *h, int32_t ncols, const double *rhs_v,
                           const double *rhs_p, const double *x0,
                           int32_t x0_cols, double *out_vp,
                           const dns_solve_opts *opts, dns_solve_stats *stats);
int dns_saddle_residual_history_col(dns_saddle *h, int32_t col, double *out,
                                    int32_t cap, int32_t *count);

/* residual history of the last solve (`krpslvprms['convstatsl']`,
 * tests/time_dep_nse_krylov.py:47); returns the number of entries written */
int dns_saddle_residual_history(dns_saddle *h, double *out, int32_t cap,
                                int32_t *count);

/* y = K x for the assembled K = [[F, JT],[J, 0]] (parity / residual checks) */
int dns_saddle_apply(dns_saddle *h, const double *x, double *y);
/* z = P^-1 r (parity checks of the preconditioner) */
int dns_saddle_apply_precond(dns_saddle *h, const double *r, double *z);
/* profiling aid: replay a hipGraph chain of `chain` identical launches of one
 * kernel of the GMRES cycle on the handle's resident data (which: 0 head,
 * 1 Fh^-1 part, 2 K apply + dots, 3 Gram-Schmidt update, 4 residual+norms,
 * 5 tail, 6 basis combine, 7 Schur block) and report microseconds per launch */
int dns_saddle_probe(dns_saddle *h, int32_t which, int32_t chain,
                     int32_t reps, double *us_per_launch);
/* eigenvalue bounds used by the Chebyshev iteration */
int dns_saddle_cheb_bounds(dns_saddle *h, double *lo, double *hi);
/* sizes of what the preconditioner keeps resident (for the published
 * per-step byte counts of bench.py): out[0..] = nnz(K), nnz(Gc) (0: recurrence
 * form), nnz(J Fh^-1) (0: triangular form), NP, fp32_store, cheb_degree,
 * schur kind, L = multigrid levels, smoothing steps, then per level l < L: n_l,
 * nnz(S_l), nnz(P_l) (0 on the coarsest).  `count` = entries available; at most `cap`
 * are written */
int dns_saddle_precond_info(dns_saddle *h, int32_t cap, int64_t *out,
                            int32_t *count);
void dns_default_precond_opts(dns_precond_opts *o);
void dns_default_solve_opts(dns_solve_opts *o);

/* ---- row-partitioned solve over several GPUs ------------------------------
 * One process per GPU.  Every rank creates the saddle system, attaches a
 * communicator and sets up the preconditioner; from then on it keeps in HBM
 * only ITS row block of every operator (K, the explicit Fh^-1, J Fh^-1) and
 * computes only those rows.  Per Arnoldi step: one halo exchange of the new
 * basis vector's velocity part (each neighbour gets exactly the entries its
 * rows reference: index lists from the column footprint), one all-gather of
 * the pressure-sized Schur input, one halo exchange of the preconditioned
 * velocity in front of the K apply, ONE all-reduce of the step's j+2 dot
 * products.  Production backend: RCCL over xGMI (grouped ncclSend/ncclRecv +
 * ncclAllReduce).  Test backend: host callbacks (torch.distributed/gloo, so
 * that two ranks can share one GPU).  Needs the explicit Fh^-1.  The
 * reference has no counterpart.
 */
typedef struct dns_comm dns_comm;
#define DNS_UNIQUE_ID_BYTES 128
/* in-place sum over ranks of `count` doubles at DEVICE pointer `dev` */
typedef int (*dns_allreduce_cb)(void *ctx, double *dev, int32_t count);
/* in-place: rank r owns dev[starts[r] .. starts[r+1]); afterwards every rank
 * holds every block; `starts` has nranks+1 entries */
typedef int (*dns_allgatherv_cb)(void *ctx, double *dev, const int32_t *starts,
                                 int32_t nranks);
/* halo exchange: send `sendcounts[q]` doubles at DEVICE `send + senddispls[q]`
 * to rank q, receive `recvcounts[q]` at DEVICE `recv + recvdispls[q]` */
typedef int (*dns_alltoallv_cb)(void *ctx, const double *send_dev,
                                const int32_t *sendcounts,
                                const int32_t *senddispls, double *recv_dev,
                                const int32_t *recvcounts,
                                const int32_t *recvdispls, int32_t nranks);
int dns_comm_unique_id(char *out_128_bytes);          /* call on rank 0     */
int dns_comm_create_rccl(int device, int32_t nranks, int32_t rank,
                         const char *unique_id_128_bytes, dns_comm **out);
int dns_comm_create_callbacks(int device, int32_t nranks, int32_t rank,
                              dns_allreduce_cb allreduce,
                              dns_allgatherv_cb allgatherv, void *ctx,
                              dns_comm **out);
/* the callback backend's halo exchange (RCCL: grouped ncclSend/ncclRecv) */
int dns_comm_set_alltoallv_cb(dns_comm *c, dns_alltoallv_cb alltoallv);
void dns_comm_destroy(dns_comm *c);
int dns_comm_stats(dns_comm *c, int64_t *n_allreduce, int64_t *n_allgather);
/* calls and bytes this rank sent: out[0..4] = all-reduce calls, all-gather-v
 * calls, halo exchanges, halo bytes, all-gather bytes */
int dns_comm_stats2(dns_comm *c, int64_t *out5);
/* Device time per collective.  While switched on, every collective that is
 * issued as a plain launch (not inside a captured graph) is bracketed by an
 * event pair on the launch stream; `dns_comm_timing` waits for the pairs and
 * returns the summed milliseconds and the number of timed calls since the
 * switch, in the order all-reduce, all-gather-v, halo exchange (the
 * ncclSend/Recv group alone, without its pack / unpack kernels).  The time
 * between the events includes the wait for the slowest peer. */
int dns_comm_set_timing(dns_comm *c, int on);
int dns_comm_timing(dns_comm *c, double *ms3, int64_t *calls3);
/* First-contact self-test, ONE primitive per call (so that a caller can say
 * which one did not come back): `which` 0 = all-reduce of `count` doubles, 1 =
 * one grouped ncclSend/ncclRecv exchange round the ring (the halo exchange's
 * call pattern), 2 = all-gather of unequal blocks (staged ncclAllGather), 3 =
 * all-gather of equal blocks (in-place ncclAllGather); `graph` 0 = plain
 * launches, 1 = captured in a hipGraph and replayed.  One call is checked
 * entry by entry (*ok), then `reps` calls are timed by an event pair on the
 * launch stream (*us_per_call).  Collective.  No reference counterpart. */
int dns_comm_selftest(dns_comm *c, int32_t which, int32_t graph, int32_t count,
                      int32_t reps, int32_t *ok, double *us_per_call);
/* out[0..2] = all-gathers issued as in-place ncclAllGather, staged
 * ncclAllGather (unequal blocks), group of ncclBroadcasts (fallback) */
int dns_comm_gather_forms(dns_comm *c, int64_t *out3);
/* 0 = in-place where the blocks are equal, staged otherwise (default); 1 = the
 * group of broadcasts; 2 = always staged (tests: one rank exercises the pack /
 * unpack path).  Environment DNS_COMM_ALLGATHER=bcast|staged sets it too. */
int dns_comm_set_gather_form(dns_comm *c, int32_t form);
/* The halo plan of a row partition, host only (no GPU needed): for the CSR
 * pattern `a` (n rows, columns < ncols_part partitioned by `col_starts`
 * [nranks+1]; columns >= ncols_part are ignored) and the rows [row0, row1) of
 * rank `rank`, the sorted distinct columns that rank references in every
 * other rank's range.  `counts` [nranks] receives the list lengths, `lists`
 * (capacity `cap`) the concatenated lists; returns the total in `*total`. */
int dns_halo_lists(const dns_csr *a, int32_t row0, int32_t row1,
                   int32_t nranks, int32_t rank, const int32_t *col_starts,
                   int32_t ncols_part, int32_t *counts, int32_t *lists,
                   int64_t cap, int64_t *total);
/* bytes of HBM the handle's matrices occupy on this rank (K, Fh^-1, J Fh^-1,
 * Schur block, F/J/JT copies): shrinks with the number of ranks */
int dns_saddle_device_bytes(dns_saddle *h, int64_t *matrix_bytes);
/* bytes of HOST memory: `kept` = the matrix copies the handle keeps (F, J, JT
 * -- whole, or the own rows of a handle created from rows --, prolongations,
 * level operators of a partitioned multigrid); `setup` = the matrices the last
 * partitioned explicit set-up had alive at its end (0 otherwise) */
int dns_saddle_host_bytes(dns_saddle *h, int64_t *kept, int64_t *setup);
/* attach before dns_saddle_setup_precond; NULL detaches */
int dns_saddle_set_comm(dns_saddle *h, dns_comm *c);
/* Rank-local construction of a row-partitioned handle: rank r of `comm` hands
 * over ITS rows of K = [[F, JT], [J, 0]] only -- `f_rows` = rows [v0, v1) of F
 * (v1 - v0 rows, `nv` columns, GLOBAL column indices), `jt_rows` = the same rows
 * of J^T (`np` columns), `j_rows` = rows [p0, p1) of J (`nv` columns), with
 * [v0, v1) = dns_partition_range(nv, ...) and [p0, p1) =
 * dns_partition_range(np, ...).  No rank ever holds a whole matrix, on the host
 * or in HBM: the rows of F and J^T a rank's part of the polynomial
 * preconditioner reaches beyond its own are fetched from their owners during
 * dns_saddle_setup_precond (degree - 1 rings of the pattern of F) and dropped
 * again; the spectral bounds come from power iterations whose product is formed
 * by rows.  The preconditioner, hence every iterate, equals bit for bit the one
 * of a handle created from whole matrices with the same communicator attached.
 * Collective (every rank of `comm` calls it).  Restrictions: explicit Fh^-1
 * (degree 2..12), dense or multigrid Schur block (its hierarchy is still built
 * from the all-gathered Schur complement); F close to symmetric (the
 * convection-dominated set-up of dns_saddle_setup_precond needs whole
 * matrices); dns_saddle_update_values takes the values of the own rows;
 * the communicator cannot be exchanged.  The reference has no counterpart (one
 * process: `lau.solve_sadpnt_smw(amat=, jmat=)`, stokes_navier_utils.py:1512);
 * this is the entry a distributed assembler binds. */
int dns_saddle_create_rows(int device, dns_comm *comm, int32_t nv, int32_t np,
                           const dns_csr *f_rows, const dns_csr *jt_rows,
                           const dns_csr *j_rows, dns_saddle **out);
/* the block partition used for n rows: [start, end) of `rank` (chunks of
 * ceil(n / nranks) rows rounded up to an even number: the two velocity dofs of
 * a node stay on one rank) */
int dns_partition_range(int32_t n, int32_t nranks, int32_t rank,
                        int32_t *start, int32_t *end);
/* raw device <-> host copies for the callback backend */
int dns_device_read(int device, const void *dev, void *host, size_t bytes);
int dns_device_write(int device, void *dev, const void *host, size_t bytes);

/* ---- device-resident IMEX time loop -------------------------------------
 * Replaces the inner loops of `time_int_utils.cnab` (tiu:104-143) and
 * `time_int_utils.sbdftwo` (tiu:320-353): the state (v, p, history) and the
 * constant system stay in HBM; per step the host supplies only what the
 * reference obtains from callbacks.
 *
 *   rhs_v = R1 (a_c v_c + a_p v_p) + cn_c*nfc_c + cn_o*nfc_o + gvec
 *   solve K [v_n; p~] = [rhs_v; rhs_p];   p_n = pscale * p~
 *
 *   CNAB : R1 = M - dt/2 A, a_c = 1,   a_p = 0,    cn_c = 3dt/2, cn_o = -dt/2
 *   SBDF2: R1 = M,          a_c = 4/3, a_p = -1/3, cn_c = 4dt/3, cn_o = -2dt/3
 */
typedef struct dns_imex_coeffs {
    double a_c, a_p;             /* weights of current / previous velocity   */
    double cn_c, cn_o;           /* weights of current / old convection      */
    double pscale;               /* p = pscale * p~  (scalep/dt, tiu:137)    */
    int32_t extrapolate_x0;      /* warm start: 0: x0 = x_c, 1: 2 x_c - x_p,
                                    2: 3 x_c - 3 x_p + x_pp (quadratic),
                                    3: 4 x_c - 6 x_p + 4 x_pp - x_ppp (cubic),
                                    4: 5, -10, 10, -5, 1 (quartic);
                                    13: cubic least-squares fit through the
                                    last five solutions (3.2, -2.8, -0.8, 2.2,
                                    -0.8: a third of the quartic's
                                    amplification of the solves' residuals) */
    int32_t carry_residual;      /* 1: the velocity residual b - K x of a step's
                                    (inexact) solve is added to the next
                                    step's right-hand side, so that the
                                    distance to the direct-solve trajectory
                                    does not grow with the number of steps
                                    (pipelined GMRES steps of dns_imex_run on
                                    one GPU below the streaming threshold;
                                    ignored elsewhere)                       */
} dns_imex_coeffs;

int dns_imex_create(dns_saddle *sys, const dns_csr *r1, dns_imex **out);
/* ... on a system created from rows (dns_saddle_create_rows): `r1_rows` = this
 * rank's rows of R1 -- rows dns_partition_range(NV, ...), NV columns, global
 * column indices; the halo lists of the stepper are exchanged instead of being
 * computed from a replicated pattern.  Collective at the first step. */
int dns_imex_create_rows(dns_saddle *sys, const dns_csr *r1_rows,
                         dns_imex **out);
void dns_imex_destroy(dns_imex *st);
/* set the state: current / previous velocity (v_p may be NULL), current p~
 * (may be NULL = 0; only seeds the warm start), the two convection history
 * vectors (NULL = 0) */
int dns_imex_set_state(dns_imex *st, const double *v_c, const double *v_p,
                       const double *ptilde_c, const double *nfc_c,
                       const double *nfc_o);
/* constant (or updated) parts of the right-hand side; NULL keeps the old */
int dns_imex_set_rhs(dns_imex *st, const double *gvec, const double *rhs_p);
/* Per-step right-hand sides known in advance -- what the reference's callbacks
 * `f_tdp(t)`, `g_tdp(t)`, `applybcs(getbcs(t))` return per step (tiu:114-127,
 * snu:1103-1126) when they depend on time only (time-dependent forcing,
 * prescribed moving-boundary data): row s of `gv` (nsteps x NV, may be NULL)
 * / `gp` (nsteps x NP, may be NULL) replaces gvec / rhs_p in the s-th step
 * after this call.  A step counter on the device selects the row, so
 * dns_imex_run replays its graphs through the whole table without the host.
 * Stepping past the last row fails with DNS_ERR_NOT_READY; dns_imex_set_rhs
 * returns to constant vectors. */
int dns_imex_set_rhs_table(dns_imex *st, int32_t nsteps, const double *gv,
                           const double *gp);
/* steps taken since the tables were uploaded / rows left (-1: no table) */
int dns_imex_table_position(dns_imex *st, int32_t *pos, int32_t *left);
/* one step; `nfc_new` (host, may be NULL) is the convection vector
 * f_vdp(v_c) evaluated by the caller at the current velocity -- it becomes
 * nfc_c, the old nfc_c becomes nfc_o (tiu:112-113).  NULL keeps both. */
int dns_imex_step(dns_imex *st, const double *nfc_new,
                  const dns_imex_coeffs *cf, const dns_solve_opts *opts,
                  dns_solve_stats *stats);
/* `nsteps` steps back to back without host callbacks (convection evaluated on
 * the device if an operator is attached, else its history stays frozen); HIP-
 * event time of the stepping loop on the handle's stream.  With GMRES and
 * `use_graph` the steps are pipelined: a fresh stepper first does up to four
 * synchronous steps (extrapolation history), then every graph the loop can ask
 * for is captured (six ring states x cycle lengths), then batches of steps are
 * replayed without host synchronisation; the ring, the predicted cycle length
 * and the graphs persist across calls, so a later call starts replaying at
 * once.  Returns DNS_NOT_CONVERGED (state advanced, outputs filled) if some
 * step ended at `maxiter`: see dns_imex_run_info. */
int dns_imex_run(dns_imex *st, int32_t nsteps, const dns_imex_coeffs *cf,
                 const dns_solve_opts *opts, dns_solve_stats *last_stats,
                 double *device_seconds, int64_t *total_iters);
/* record of the last dns_imex_run: steps that did not converge and the first
 * of them (index within the run, -1: none), steps that were repeated because
 * a batch mispredicted its cycle length, graphs captured inside the call */
int dns_imex_run_info(dns_imex *st, int32_t *unconverged, int32_t *first_bad,
                      int32_t *replayed, int32_t *captures);
/* diagnostics of the row-partitioned step (tests): out[0] = steps built
 * (launched or captured; graph replays are not counted), out[1] = of them the
 * steps whose one-step cycle evaluated the convection cells of the new
 * velocity in its tail, out[2] = the steps that left the cell kernel out
 * because the tail before had run it */
int dns_imex_step_counters(dns_imex *st, int64_t *out3);
/* v (NV) and p = pscale*p~ (NP) of the current state */
int dns_imex_get_state(dns_imex *st, double *v, double *p);
/* ||v||_2 of the current velocity (blow-up guard, tiu:94-103) */
int dns_imex_vnorm(dns_imex *st, double *out);

/* ---- convection on the device --------------------------
 * N(u)u = inner(grad(u)*u, v)*dx for P2 velocities on triangles: replaces the
 * host callback `f_vdp` = `get_v_conv_conts(semi_explicit=True)` ->
 * `dolfin_to_sparrays.get_convvec` (stokes_navier_utils.py:1136-1140,103-107;
 * dolfin_to_sparrays.py:427-472) including its `append_bcs_vec`
 * (dolfin_to_sparrays.py:49-64).  Mesh data as the caller's FE library has it:
 *   cell_vdofs[12*c + 2*a + i] : FULL-space velocity dof of local node a,
 *        component i; local nodes 0..2 = vertices, 3..5 = midpoints of the
 *        edges opposite to vertex 0, 1, 2
 *   glam[6*c + 2*k + d]  : d-th component of grad(lambda_k) on cell c
 *   area[c]; invinds (inner dofs, order of the condensed vectors);
 *   dbcinds / dbcvals (Dirichlet dofs and values)
 */
typedef struct dns_conv dns_conv;
int dns_conv_create_p2(int device, int32_t ncells, const int32_t *cell_vdofs,
                       const double *glam, const double *area, int32_t vdim,
                       int32_t nv_inner, const int32_t *invinds, int32_t ndbc,
                       const int32_t *dbcinds, const double *dbcvals,
                       dns_conv **out);
void dns_conv_destroy(dns_conv *cv);
int dns_conv_set_dbcvals(dns_conv *cv, const double *dbcvals);
/* Dirichlet values that change from step to step (`append_bcs_vec` with
 * controlled boundary values, snu:1003-1006,1152-1157): `nrows` value sets
 * (nrows x ndbc).  Attached to an IMEX stepper, row s is used by the s-th step
 * after the stepper's tables were (re)set (the values at the step's CURRENT
 * time, where N(v_c)v_c is evaluated); in the trapezoidal sweeps the row is the
 * trajectory slot.  dns_conv_set_dbcvals returns to one constant set. */
int dns_conv_set_dbc_table(dns_conv *cv, int32_t nrows, const double *dbcvals);
/* the row the host-driven entry points (dns_conv_apply / _assemble) use */
int dns_conv_set_dbc_row(dns_conv *cv, int32_t row);
/* out = scale * N(u)u restricted to the inner dofs (host in/out; parity) */
int dns_conv_apply(dns_conv *cv, const double *v_inner, double scale,
                   double *out);
/* let the stepper evaluate nfc_c = scale * N(v_c)v_c itself every step
 * (scale = -1: "goes to the rhs", stokes_navier_utils.py:1128-1140); NULL
 * detaches.  dns_imex_step then takes nfc_new = NULL and dns_imex_run
 * advances the convection history instead of freezing it */
int dns_imex_set_convection(dns_imex *st, dns_conv *cv, double scale);

/* ---- standalone kernels (parity tests, micro-benchmarks) -----------------
 * upload, run the same device kernels the solver uses, download */
int dns_spmv(int device, const dns_csr *a, const double *x, double *y,
             double alpha, double beta, int32_t variant);
int dns_dot(int device, int64_t n, const double *x, const double *y,
            double *out);
int dns_axpy(int device, int64_t n, double a, const double *x, double *y);
int dns_gemv(int device, int32_t n, const double *a_rowmajor, const double *x,
             double *y, double alpha);
/* dense in-place inverse by the device Gauss-Jordan used for the Schur block */
int dns_dense_inverse(int device, int32_t n, double *a_rowmajor);

/* repeat y = A x `reps` times on resident data; reports the average kernel
 * time (HIP events on the launch stream) -- the roofline measurement */
int dns_spmv_bench(int device, const dns_csr *a, int32_t variant, int32_t reps,
                   int32_t warmup, double *avg_seconds, double *checksum);
/* y = K x for K = [[F, JT],[J, 0]] (CSR, velocity rows/columns first, `nv` of
 * them) through the 2x2-blocked PAIR format: rows and columns are taken in
 * pairs (the two velocity dofs of a node; consecutive pressure dofs), an entry
 * is a 2x2 block = four values + one 16-bit column offset, served by one
 * 16-byte gather of x.  Holds any K with even nv and even size (explicit zeros
 * where a block is not full); DNS_ERR_BAD_ARGUMENT otherwise.  reps > 0 and
 * avg_seconds != NULL: seconds per launch by HIP events, as dns_spmv_bench;
 * format_bytes: device bytes of the format (may be NULL).  Inside a solver
 * the format carries the K applies of the bandwidth regime automatically
 * (DNS_PAIR=0 turns it off). */
int dns_spmv_pair(int device, const dns_csr *k, int32_t nv, const double *x,
                  double *y, int32_t reps, int32_t warmup, double *avg_seconds,
                  int64_t *format_bytes);

/* Multigrid Schur block for pressure spaces too large for the dense inverse
 * (refined meshes).  `prol[l]` (CSR, n_l x n_{l+1}, l = 0 .. nprol-1) are the
 * prolongations of a nested hierarchy of pressure spaces, finest first (n_0 =
 * NP); the coarsest space gets a dense inverse, so keep n_nprol at a few
 * thousand.  At `dns_saddle_setup_precond(schur = DNS_SCHUR_MG)` the library
 * forms the sparse Schur complement S_0 = J Fh^-1 JT (explicit Fh^-1) or
 * J D^-1 JT (recurrence), the Galerkin operators S_{l+1} = P_l^T S_l P_l and
 * applies Sh^-1 ~ one V(nu,nu)-cycle with damped Jacobi smoothing (in the
 * row-partitioned solve every rank runs the whole cycle).
 * (Reference counterpart: none -- its direct solver does not need one.) */
int dns_saddle_set_schur_mg(dns_saddle *h, int32_t nprol, const dns_csr *prol,
                            int32_t smooth_steps);
/* Tuning knobs of ONE handle, to be set before dns_saddle_setup_precond (the
 * environment variables DNS_<NAME> give the defaults when the handle is
 * created; this call replaces process-global state in tests and drivers):
 *   "stream_nnz"    matrices with at least this many non-zeros take the
 *                   LDS-streaming kernels (bandwidth regime)
 *   "pair"          0/1: pair format of K in the bandwidth regime
 *   "mg_dense_max"  first multigrid level <= this gets the dense inverse
 *   "mg_dense_half_max"  ... and a level up to this size gets it in half
 *                   precision (one bandwidth-bound launch instead of the five
 *                   latency-bound launches of a sparse level); 0 = never
 *   "mg_part_min"   levels with at least this many rows are row-partitioned
 *   "mg_fused"      0/1: fused V(2,2) operators
 *   "mg_cheb"       0/1: Chebyshev pair of smoothing weights; "mg_cheb_alpha"
 *   "mg_cycles"     1 or 2 cycles per application; "mg_rho"
 *   "dist_graph"    0/1: hipGraph replay with captured RCCL calls
 *   "cycle_first"   > 0: Krylov steps in the first cycle of every solve
 *                   (default: what the previous solve needed, plus one)
 *   "dist_lazy1"    0/1: one-step cycles of a partitioned solve leave the
 *                   first basis vector un-normalised; ||r||, ||b|| travel with
 *                   the step's dots (one all-reduce per time step less)
 *   "dist_x0_exchange"  0/1: halo exchange of the start vector at the head of
 *                   every cycle of a partitioned solve (default 0: it is valid)
 *   "part_setup"    0/1: with a communicator, every rank forms only the rows
 *                   of Fh^-1, J Fh^-1 and of the Schur complement its blocks
 *                   are made of (default 1; 0: every rank forms all rows)
 *   "mg_stream_nnz": operators of the multigrid cycle with at least this
 *                   many non-zeros run as fp32 streams through the
 *                   LDS-streaming kernels (default 600000; "stream_nnz",
 *                   1300000, is the threshold of K, Fh^-1 and J Fh^-1)
 *   "oversolve"   : 1 / 0 = the solves of a pipelined batch run the columns
 *                   of their cycle instead of stopping at the tolerance / stop
 *                   at it (slack-column policy); -1 = the default (on with the
 *                   multigrid Schur block below 1.5e6 unknowns)
 *   "oversolve_cmin": shortest cycle the oversolve policy tries (default 2
 *                   with the multigrid block, else 1)
 * Unknown names: DNS_ERR_BAD_ARGUMENT.  (No reference counterpart.) */
int dns_saddle_set_option(dns_saddle *h, const char *name, double value);

/* ---- linearised convection matrices + Newton/Picard trapezoidal sweeps ------
 * (reference: `get_v_conv_conts` snu:109-133 with `get_convmats` dts:325-376
 * and `condense_velmatsbybcs` dts:610-642; `_get_mats_rhs_ts` snu:1016-1047;
 * the time loop snu:1402-1566)
 *
 * dns_conv_bind_pattern: the CSR pattern (values ignored) the condensed
 *   N1(u) (+ N2(u)) are assembled into -- it must contain theirs, e.g. the
 *   pattern of A (symmetric-gradient form) or of M + A kept with its zeros.
 * dns_conv_assemble (host pointers; the parity-test entry): values of
 *   N1(u) [newton = 0] or N1(u) + N2(u) [newton = 1] in that pattern,
 *   rhsbc = -N[:, Dirichlet cols] * Dirichlet values (may be NULL) and
 *   rhscon = N(u)u (may be NULL).
 */
int dns_conv_bind_pattern(dns_conv *cv, const dns_csr *pattern);
int dns_conv_assemble(dns_conv *cv, const double *u_inner, int32_t newton,
                      double *nvals, double *rhsbc, double *rhscon);

/* the same with two sets of Dirichlet values: `dbcvals_lin` belong to the
 * linearisation field (they enter N1, N2 and N(u)u), `dbcvals_rhs` multiply
 * the Dirichlet columns in `rhsbc` (NULL: the same set) -- the steady Picard
 * iteration uses the old values for the field and the new ones for the rhs
 * (snu:446-455); the operator's own value set / table is not touched */
int dns_conv_assemble2(dns_conv *cv, const double *u_inner,
                       const double *dbcvals_lin, const double *dbcvals_rhs,
                       int32_t newton, double *nvals, double *rhsbc,
                       double *rhscon);

/* Trapezoidal stepper: M, A given as value arrays in the pattern of `sys`'s F
 * block (to which `conv` must be bound); two trajectory buffers of `nslots`
 * velocities each hold the linearisation points of the running sweep
 * (`which`) and the velocities it produces (`1 - which`).
 *   start : v_c = iniv, f_c - N_c v_c evaluated there, update norm reset
 *   step  : one step of size dt linearised about traj[lin_which][lin_slot];
 *           F = M + dt/2 (A + N_n) is formed on the device inside `sys`
 *           (same pattern, preconditioner kept); result -> traj[1-lin_which]
 *           [out_slot] (out_slot < 0: not stored).  `extrapolate_x0` = order
 *           of the warm start from the last solutions, 0 .. 4 as in
 *           dns_imex_coeffs (the reference's `krylovini='upd'` is order 1,
 *           snu:1493-1503).  The part of the right-hand side that depends on
 *           the current velocity through the convection, f_c - N_c v_c, is
 *           evaluated as fv - N(v_c)v_c (N1(u)u = N2(u)u = N(u)u): no second
 *           matrix assembly per step
 *   get_state: v_c and p = -p~/dt (snu:1542)
 *   update_norm: sum of dt ||v_n - v_lin||_M^2 since `start` (snu:1557-1560)
 */
typedef struct dns_trap dns_trap;
int dns_trap_create(dns_saddle *sys, dns_conv *conv, const double *m_vals,
                    const double *a_vals, int32_t nslots, dns_trap **out);
void dns_trap_destroy(dns_trap *t);
int dns_trap_set_rhs(dns_trap *t, const double *fv, const double *fp);
int dns_trap_traj_write(dns_trap *t, int32_t which, int32_t slot,
                        const double *v);
int dns_trap_traj_read(dns_trap *t, int32_t which, int32_t slot, double *v);
/* Asynchronous writer of the trajectory store (replaces the
 * blocking per-step `dou.save_npa`, reference snu:1012-1014, 1424-1431): the
 * copy of the slots [slot0, slot0+count) of traj[which] (count*NV doubles,
 * into a page-locked buffer of the library's) is queued on a copy stream of
 * its own
 * behind the work the solver's stream has been given so far; the call returns
 * at once and the next sweep runs while the data travels.  The solver's stream
 * waits for a pending export only before it overwrites traj[which].
 * dns_trap_traj_export_wait blocks until all exports have arrived and fills
 * the `host` arrays (which have to be alive until then); any host thread may
 * call it. */
int dns_trap_traj_export_async(dns_trap *t, int32_t which, int32_t slot0,
                               int32_t count, double *host);
int dns_trap_traj_export_wait(dns_trap *t);
int dns_trap_start(dns_trap *t, const double *iniv, int32_t newton);
int dns_trap_step(dns_trap *t, double dt, int32_t lin_which, int32_t lin_slot,
                  int32_t out_slot, int32_t newton, int32_t extrapolate_x0,
                  const dns_solve_opts *opts, dns_solve_stats *stats);
int dns_trap_get_state(dns_trap *t, double *v, double *p);
int dns_trap_update_norm(dns_trap *t, double *out);
/* On a row-partitioned system (dns_saddle_set_comm) the stepper's assembly is
 * partitioned like the solve and no solution is gathered per step: trajectory
 * slots and states hold a rank's rows and its halo.  dns_trap_traj_read,
 * dns_trap_traj_export_async, dns_trap_get_state and dns_trap_update_norm are
 * COLLECTIVE then (they gather / all-reduce first): every rank calls them. */
/* time-dependent data of the sweeps, one row per trajectory slot (= time
 * instance; nslots rows each, NULL = not time dependent): `fv_tab` replaces fv
 * (forcing `fvtd(t)` and the stiffness contribution of controlled boundary
 * values, snu:1466-1468), `fp_tab` replaces fp, `mbc_tab` = M[:, cnt] bcvals(t)
 * adds `mbcs_n - mbcs_c` to the rhs (snu:1044-1045, 1438-1441) */
int dns_trap_set_tables(dns_trap *t, const double *fv_tab, const double *fp_tab,
                        const double *mbc_tab);
/* one step with the low-rank feedback terms of `_get_mats_rhs_ts`
 * (snu:1036-1042, passed at snu:1505-1512): system matrix F - dt/2 U V_n
 * (Sherman-Morrison-Woodbury: r more solves with the same K), right-hand side
 * + dt/2 U (V_c v_c).  umat: NV x r column major; vmat_c (may be NULL),
 * vmat_n: r x NV row major; synchronous steps only */
int dns_trap_step_fb(dns_trap *t, double dt, int32_t lin_which,
                     int32_t lin_slot, int32_t out_slot, int32_t newton,
                     int32_t extrapolate_x0, const dns_solve_opts *opts,
                     dns_solve_stats *stats, int32_t r, const double *umat,
                     const double *vmat_c, const double *vmat_n);
/* pipelined sweeps: with cycle_len > 0 `dns_trap_step` only enqueues (one GMRES
 * cycle of that length, no host synchronisation, `stats` not filled); the
 * device counts solves / unconverged solves / Krylov steps and keeps the update
 * norm; `dns_trap_poll` synchronises, hands the counters back (and resets
 * them) and folds the device part of the update norm in.  A sweep with
 * `fails > 0` has to be repeated with cycle_len = 0. */
int dns_trap_set_pipeline(dns_trap *t, int32_t cycle_len);
/* `count` pipelined steps of step size dt in a row: step k has its
 * linearisation point in slot slot0 + k of traj[lin_which] and stores its
 * velocity in the same slot of the other buffer (the time loop of a sweep,
 * snu:1402-1566, without a trip through the caller per step).  Needs
 * dns_trap_set_pipeline(cycle_len > 0) */
int dns_trap_run(dns_trap *t, double dt, int32_t lin_which, int32_t slot0,
                 int32_t count, int32_t newton, int32_t extrapolate_x0,
                 const dns_solve_opts *opts);
int dns_trap_poll(dns_trap *t, int32_t *solves, int32_t *fails, int32_t *iters,
                  int32_t *maxit);
/* Oversolve of the pipelined batches: with stop_frac > 0 a solve does not stop
 * at the tolerance but runs the columns of its cycle (the nodes of a replayed
 * cycle are paid for either way) until its residual is below stop_frac x
 * tolerance; 0 (default) = stop at the tolerance.  dns_trap_poll_ext: the
 * counters of dns_trap_poll in out6[0..3], out6[4] = the largest number of
 * columns a solve NEEDED to meet the tolerance, out6[5] = the sum of those
 * numbers over the batch's solves, out2[0] = batch maximum of
 * final residual / tolerance, out2[1] = ... of the residual in front of the
 * last column / tolerance: what the host's choice of the next cycle length
 * goes by. */
int dns_trap_set_oversolve(dns_trap *t, double stop_frac);
int dns_trap_poll_ext(dns_trap *t, int32_t *out6, double *out2);
/* checkpoint / restore of the stepper's state between pipelined batches: a
 * batch in which a step did not converge within the agreed cycle length is
 * repeated from the checkpoint (call `dns_trap_checkpoint` behind a
 * `dns_trap_poll`; `dns_trap_restore` leaves the pipeline switched off and
 * re-assembles N_c / f_c at the restored velocity) */
int dns_trap_checkpoint(dns_trap *t);
int dns_trap_restore(dns_trap *t, int32_t newton);

/* ---- resident helpers of the caller side of the path -----------------------
 * A CSR matrix kept in HBM for repeated products with host vectors: what the
 * closures of `solve_nse` do per step with SciPy -- `applybcs` (snu:1111-1115:
 * -A[:, cnt] vals, -J[:, cnt] vals, M[:, cnt] vals),
 * `condense_velmatsbybcs(get_rhs_only=True)` (dts:610-630) -- and the J / J^T
 * products of the decoupled pressure solve (snu:1622-1628).
 *   y = alpha A x + beta y   (host vectors) */
typedef struct dns_op dns_op;
int dns_op_create(int device, const dns_csr *a, dns_op **out);
void dns_op_destroy(dns_op *op);
int dns_op_apply(dns_op *op, const double *x, double *y, double alpha,
                 double beta);
/* `append_bcs_vec` (dts:49-64): out = NaN everywhere, then out[invinds] =
 * v_inner, then out[bcinds] = bcvals (boundary values win; of a repeated index
 * the last occurrence wins, as in NumPy's fancy assignment) */
typedef struct dns_bcmap dns_bcmap;
int dns_bcmap_create(int device, int32_t vdim, int32_t ninv,
                     const int32_t *invinds, int32_t nbc, const int32_t *bcinds,
                     dns_bcmap **out);
void dns_bcmap_destroy(dns_bcmap *m);
int dns_bc_scatter(dns_bcmap *m, const double *v_inner, const double *bcvals,
                   double *out_full);

/* attainable HBM bandwidth of the device, measured with plain streaming
 * kernels over `bytes` of fp64 data (kind 0: read + reduce, 1: copy,
 * 2: triad a = b + s c); reports GB/s of the bytes the kernel moves.  The SpMV
 * roofline fraction is quoted against the 8 TB/s peak AND against this. */
int dns_hbm_probe(int device, int64_t bytes, int32_t kind, int32_t reps,
                  double *gbytes_per_s);

#ifdef __cplusplus
}
#endif
#endif /* DNS_AMD_H */
