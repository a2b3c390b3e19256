/*
 * include/sbgpu.h -- C ABI of libsbgpu.so, the MI355X (gfx950) implementation of
 * Strawberry's per-locus EM hot path.
 *
 * The reference (ruolin/strawberry v1.1.2, /root/reference) has no plugin / FFI
 * boundary; the narrowest seam is the EmSolver class and its single call site:
 *
 *     EmSolver em;                                   include/estimate.hpp:230-257
 *     success = em.init(niso, n, alpha);             src/estimate.cpp:305-307
 *     if (success) em.run();                         src/estimate.cpp:308
 *     ... em._theta[i] ...                           src/estimate.cpp:310-329
 *
 * One call per locus would serialise the GPU, so the boundary is BATCHED: the
 * host collects the (n, alpha) of many loci into a CSR-of-loci arena, one call
 * solves them all, and the per-locus result is (theta, status, iters).  The
 * per-locus EmSolver-shaped adapter lives above this ABI (include/sbgpu_host.hpp in
 * C++14, strawberry_amd/em.py for the Python harness).
 *
 * Around that seam the same batching covers what feeds it: read pairs ->
 * unique hits (sbgpu_collapse_pairs_host), hit x isoform compatibility and exon bins
 * (sbgpu_exonbin_*, sbgpu_bins_create*), bin weights (sbgpu_binweight_*), and what follows it:
 * FPKM / Frac / TPM (sbgpu_abundance_device, sbgpu_tpm_device) and the reference's two output
 * formats (sbgpu_format_*).  sbgpu_quantify_host chains them in one call.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on
 * success and a negative SBGPU_E* code on failure (sbgpu_last_error() has the
 * text).  Per-locus outcomes are NOT errors, they are `status` values that
 * mirror the reference's bool returns.  "d_" pointers are device (HBM)
 * addresses, everything else is host memory.  `stream` is a hipStream_t passed
 * as void* (NULL = HIP's null stream, as everywhere in HIP).  A context is used from one host
 * thread at a time (one process per GPU).
 */
#ifndef SBGPU_H_
#define SBGPU_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SBGPU_VERSION_MAJOR 0
#define SBGPU_VERSION_MINOR 1

/* ---- function return codes ------------------------------------------------ */
#define SBGPU_OK 0
#define SBGPU_EINVAL (-1)   /* bad argument / malformed batch                    */
#define SBGPU_EHIP (-2)     /* a HIP runtime call failed                          */
#define SBGPU_ENODEV (-3)   /* no gfx950 device / device index out of range       */
#define SBGPU_ENOMEM (-4)   /* host or device allocation failed                   */
#define SBGPU_ESHAPE (-5)   /* a locus exceeds the supported shape (niso > 512)   */
#define SBGPU_EUNSUPPORTED (-6) /* the device form does not cover this input: use the host form (see the call) */
#define SBGPU_ERCCL (-7)    /* RCCL could not be loaded or one of its calls failed    */

/* ---- per-locus status: the reference's two bools, src/estimate.cpp:305-308 - */
#define SBGPU_EM_OK 0          /* init()==true,  run()==true, converged (estimate.cpp:480 break) */
#define SBGPU_EM_INIT_EMPTY 1  /* init()==false: no row has a weight > 1e-5 (estimate.cpp:391);
                                  theta = theta0; the caller drops the locus (alignments.cpp:1524) */
#define SBGPU_EM_DENOM_ZERO 2  /* run()==false: a row denominator was 0 (estimate.cpp:451-453);
                                  theta = theta0; the caller proceeds (bool ignored, estimate.cpp:308) */
#define SBGPU_EM_MAXITER 3     /* run()==true after all 1000 iterations (estimate.hpp:237)         */
#define SBGPU_EM_UNSOLVED (-1) /* no kernel wrote a result for the locus: what sbgpu_em_run_device sets every status
                                  to before it launches; only seen together with an SBGPU_EHIP from
                                  sbgpu_synchronize / sbgpu_em_batch / sbgpu_quantify_host (a wide-locus
                                  barrier timed out)                                                */

/* EmSolver constants, include/estimate.hpp:237,241 and src/estimate.cpp:380 */
#define SBGPU_EM_MAX_ITER 1000
#define SBGPU_EM_THETA_CHANGE_LIMIT 1e-2
#define SBGPU_EM_ROW_EPS 1e-5

typedef struct sbgpu_ctx sbgpu_ctx_t;   /* one per GPU / process */
typedef struct sbgpu_plan sbgpu_plan_t; /* shapes of one batch + its size-class schedule */

/*
 * A batch of loci in CSR-of-loci form = the arguments of EmSolver::init
 * (src/estimate.cpp:366-368: num_iso, count, model) for n_loci loci:
 *   locus l has nrow_l = row_off[l+1]-row_off[l] exon bins (rows) and
 *   niso_l = iso_off[l+1]-iso_off[l] isoforms (columns);
 *   count[row_off[l] + i]               = n_i        (vector<int> count)
 *   F[f_off[l] + i*niso_l + j]          = alpha[i][j] (vector<vector<double>> model)
 *   f_off[l+1]-f_off[l] must equal nrow_l*niso_l.
 */
typedef struct {
   int64_t n_loci;
   const int64_t *row_off; /* [n_loci+1] */
   const int64_t *iso_off; /* [n_loci+1] */
   const int64_t *f_off;   /* [n_loci+1] */
   const int32_t *count;   /* [row_off[n_loci]] */
   const double *F;        /* [f_off[n_loci]]   */
} sbgpu_batch_t;

/* Globals the reference's epilogue reads (include/common.h:25-85), passed
 * explicitly: no globals cross the ABI.                                         */
typedef struct {
   int32_t total_mapped_reads;   /* Sample::total_mapped_reads(), alignments.cpp:1372      */
   int32_t effective_len_norm;   /* common.cpp default false; estimate.cpp:317             */
   int32_t filter_by_expression; /* common.cpp default true;  estimate.cpp:346             */
   int32_t reserved;
   double insert_mean;           /* InsertSize::_mean, only read when effective_len_norm   */
   double min_isoform_frac;      /* kMinIsoformFrac: 0.01, or 0 under -r (Strawberry.cpp:158-162) */
} sbgpu_abundance_params_t;

/* ---- library / context ------------------------------------------------------ */
const char *sbgpu_version(void);
const char *sbgpu_last_error(void);
/* 16 hex digits: a hash of every source file the library was built from (csrc/Makefile).  Profiles name the build
 * they were taken with; a reader (bench.py) does not quote them for another build.                              */
const char *sbgpu_build_id(void);
int sbgpu_device_count(void);

/* Bind to HIP device `device`; creates the context's streams and workspace.
 * Concurrency.  A context (and every plan / bins handle made with it) serves ONE call at a time, from one host thread:
 * the multi-stage entry points (sbgpu_quantify_*, sbgpu_bins_create_device, sbgpu_binweight_device,
 * sbgpu_collapse_pairs_device, sbgpu_em_batch) share per-context device state -- grow-only scratch arenas, the
 * insert-size table's support, a plan's exchange buffers and epoch -- so two of them in flight on one context, even on
 * different streams, race.  Use one context per concurrent caller (contexts are cheap; one process per GPU uses one).
 * Device inputs handed to sbgpu_quantify_device / sbgpu_collapse_pairs_device must be complete when the call is made
 * (synchronise the stream that produced them): those entry points run on the context's own stream.                */
int sbgpu_init(int device, sbgpu_ctx_t **ctx_out);
int sbgpu_finalize(sbgpu_ctx_t *ctx);
/* Device memory the library keeps between calls: the arenas of handles and plans go back to a process-wide pool when they are
 * released (a sample-sized call's arenas are tens of GB; allocating and freeing them per call costs seconds), at most
 * half of the device's memory (environment: SBGPU_POOL_GB).  A caller that wants that memory for something else
 * hands it back to the driver with this; returns the bytes released.                                                  */
int64_t sbgpu_release_idle_memory(void);
/* Device facts for roofline accounting: out[0]=#CUs, out[1]=wave size,
 * out[2]=LDS bytes per CU, out[3]=max clock kHz, out[4]=total HBM bytes (MiB).   */
int sbgpu_device_info(sbgpu_ctx_t *ctx, int64_t out[8]);
int sbgpu_synchronize(sbgpu_ctx_t *ctx, void *stream);

/* ---- plan: shapes -> size classes --------------------------------------------
 * Reads only the three HOST offset arrays.  Sorts loci into (columns, rows per
 * lane, lanes per locus) size classes, orders each class by decreasing work and
 * uploads offsets + class lists.  A plan is reusable for any (count, F) of the
 * same shapes (e.g. every bench step).                                          */
int sbgpu_plan_create(sbgpu_ctx_t *ctx, int64_t n_loci, const int64_t *row_off,
                      const int64_t *iso_off, const int64_t *f_off,
                      sbgpu_plan_t **plan_out);
int sbgpu_plan_destroy(sbgpu_plan_t *plan);
/* out[0]=n_loci, out[1]=total rows, out[2]=total isoforms, out[3]=total F elements,
 * out[4]=#size classes in use, out[5]=#loci on the streaming (large-shape) path,
 * out[6]=algorithmic bytes of the batch (SURVEY 8(d) B_locus summed, fp64).       */
int sbgpu_plan_info(const sbgpu_plan_t *plan, int64_t out[8]);
/* Per-class description for profiling: fills up to `cap` rows of 6 int64:
 * {kind (0 tile,1 stream), C, R, G, n_loci, n_waves}.  Returns #classes.          */
int sbgpu_plan_classes(const sbgpu_plan_t *plan, int64_t *out, int cap);

/* Which kernel serves each locus: out[l] = 0/1/2 wave kinds (half/base/double tile),
 * 3 block, 4 tall block, 5 streaming kind (profiling / roofline accounting only).  */
int sbgpu_plan_locus_kinds(const sbgpu_plan_t *plan, int8_t *out);

/* ---- the hot path: EmSolver::init + run for every locus of the plan ----------
 * Replaces src/estimate.cpp:305-308 (EmSolver::init :366-409, ::run :411-488).
 * Inputs resident in HBM; outputs written to HBM:
 *   d_theta[iso_off[n_loci]]  = em._theta of every locus (concatenated)
 *   d_status[n_loci]          = SBGPU_EM_*
 *   d_iters[n_loci]           = E-steps started (0 for INIT_EMPTY).  The reference's iteration count for status
 *                               OK and MAXITER.  UNSPECIFIED (+-1) for SBGPU_EM_DENOM_ZERO: the count of a failed
 *                               solve is not part of the reference's output, and WHEN a decaying denominator is
 *                               flushed to zero depends on one rounding -- the E-step's dot product is a chain of
 *                               fused multiply-adds (csrc/em_device.h: `sum = fma(F, theta, sum)`), in which a product
 *                               that is subnormal by itself survives, where the reference's separate multiply
 *                               (-Ofast: FTZ) flushes it: one locus in 6e5 of the stress runs sees its zero one
 *                               iteration later (theta and status identical; profiles/r05_stress.txt)
 * Asynchronous on `stream`.                                                     */
int sbgpu_em_run_device(sbgpu_ctx_t *ctx, const sbgpu_plan_t *plan,
                        const int32_t *d_count, const double *d_F,
                        double *d_theta, int32_t *d_status, int32_t *d_iters,
                        void *stream);

/* The same for a caller that runs batch after batch: the kernels start behind `fork_stream`'s work (the inputs) and their
 * completion is joined into `join_stream` -- the stream the caller's epilogue for THIS batch runs on -- instead of into the
 * stream they started from.  `fork_stream` is then free at once: the next call's kernels queue behind this call's on the
 * library's own streams, kind by kind, without a cross-stream hand-off between two batches (the hand-offs are ~70 us of a
 * 0.84 ms C3 step).  Two consecutive calls' kernels may run side by side (the kinds that end last alternate between two of
 * the library's streams), call i + 2's run behind call i's: the caller keeps consecutive batches' outputs apart -- theta /
 * status / iterations of call i + 1 in other buffers than call i's -- and lets `fork_stream` wait for the epilogue that
 * last read a buffer set before it hands the set to a new call.                                                          */
int sbgpu_em_run_device_split(sbgpu_ctx_t *ctx, const sbgpu_plan_t *plan, const int32_t *d_count, const double *d_F,
                              double *d_theta, int32_t *d_status, int32_t *d_iters, void *fork_stream,
                              void *join_stream);

/* The fp32 variant (BASELINE config 5, "fp32 vs fp64 tolerance sweep"): the same kernels with F, theta and all
 * arithmetic in fp32 (d_F, d_theta are float arrays; fp32 denormals flush too).  It exists to measure what
 * fp32 costs in accuracy and buys in speed; it is NOT a parity path: results differ from the reference's fp64
 * EmSolver (tools/c5_sweep.py).  Loci of up to 64 isoforms only (else SBGPU_EUNSUPPORTED).        */
int sbgpu_em_run_device_f32(sbgpu_ctx_t *ctx, const sbgpu_plan_t *plan,
                            const int32_t *d_count, const float *d_F,
                            float *d_theta, int32_t *d_status, int32_t *d_iters,
                            void *stream);

/* BASELINE config 5, "bias kernel fused into the E-step": the weight of (bin i, isoform j) is
 * F_ij * 2^(row_bias[i] * iso_bias[j]); the factor is applied to the operand while the register tile is loaded (once
 * per solve), the iterations then run on the biased tile.  d_row_bias[total rows], d_iso_bias[total isoforms], both in
 * [-1, 1] (b_ij in [0.5, 2]); how they are derived is the caller's model -- the reference has no bias arithmetic
 * (src/bias.cpp is comments), so this is NOT a parity path; bench.py --workload c5 derives row_bias from the bins'
 * GC ratio as sbgpu_binseq_device measures it.  Same result as sbgpu_em_run_device on the pre-multiplied weights up
 * to exp2's rounding.  Served by the tile kernels and the multi-workgroup kernel of the wide loci (up to 512 isoforms);
 * a plan with phases, or with a locus on the streaming fallback (more than 512 isoforms or 256 workgroups): SBGPU_EUNSUPPORTED.
 * The fp32 form covers loci of up to 64 isoforms only (it is a tolerance experiment, not a product path).              */
int sbgpu_em_run_device_bias(sbgpu_ctx_t *ctx, const sbgpu_plan_t *plan, const int32_t *d_count, const double *d_F,
                             const double *d_row_bias, const double *d_iso_bias, double *d_theta, int32_t *d_status,
                             int32_t *d_iters, void *stream);
int sbgpu_em_run_device_bias_f32(sbgpu_ctx_t *ctx, const sbgpu_plan_t *plan, const int32_t *d_count, const float *d_F,
                                 const float *d_row_bias, const float *d_iso_bias, float *d_theta, int32_t *d_status,
                                 int32_t *d_iters, void *stream);

/* Timing events around the EM kernels are off by default (they cost a few microseconds
 * per call); sbgpu_set_timing(ctx, 1) turns them on for the calls that follow.       */
int sbgpu_set_timing(sbgpu_ctx_t *ctx, int on);
/* With timing on, sbgpu_quantify_host / _device bracket each of their kernel stages (exon bins, grouping, packing,
 * pairs, bin weights, EM) with HIP events on the stream the kernels run on.  Reads the stages of the last such call
 * (waits for them): fills ms[i] / names[i] (static strings) for up to `cap` stages, returns their number or a
 * negative SBGPU_E* code.  The times are the kernels' own: host work between the stages is not in them.       */
int sbgpu_last_stage_ms(sbgpu_ctx_t *ctx, int cap, float *ms, const char **names);

/* Device time of the last sbgpu_em_run_device per kernel kind (HIP events recorded
 * on the stream each kind was launched on, all phases of the kind included):
 * ms[6] = {wave half tile, wave base tile, wave double tile, block, tall block,
 * stream}, 0 for kinds not launched or when timing is off.  Synchronises with those
 * events.                                                                          */
int sbgpu_em_last_kernel_ms(sbgpu_ctx_t *ctx, float ms[6]);

/* The wave kind runs in phases: phase 0 takes every locus up to a first iteration
 * limit, each later phase continues the loci still running in layouts with more lanes
 * per locus.  Fills ms[0 .. n) with the device time of each phase of the
 * last sbgpu_em_run_device (timing on) and returns n (0: one phase or timing off).   */
int sbgpu_em_last_phase_ms(sbgpu_ctx_t *ctx, float *ms, int cap);

/* Host-buffer convenience form (what a cgo/JNI/ctypes or the C++ driver binds
 * first): plans, uploads, solves, downloads, synchronises.                      */
int sbgpu_em_batch(sbgpu_ctx_t *ctx, const sbgpu_batch_t *host_batch,
                   double *theta_out, int32_t *status_out, int32_t *iters_out);

/* ---- abundance epilogue: theta -> FPKM / Frac / keep -------------------------
 * Replaces LocusContext::estimate_abundances, src/estimate.cpp:314-355.
 *   d_length[n_iso]  exonic length L_j of every isoform (estimate.hpp:98)
 *   d_fpkm, d_frac   [n_iso] doubles
 *   d_keep[n_iso]    0 erased (Frac < min_isoform_frac, or locus INIT_EMPTY),
 *                    1 kept, 2 kept but "NA" (effective length < 0)
 *   d_sum_fpkm[1]    = sum of FPKM over kept isoforms of this rank (the operand
 *                    of the TPM all-reduce, alignments.cpp:1821-1824); overwritten. */
int sbgpu_abundance_device(sbgpu_ctx_t *ctx, const sbgpu_plan_t *plan,
                           const double *d_theta, const int32_t *d_status,
                           const int32_t *d_length,
                           const sbgpu_abundance_params_t *params,
                           double *d_fpkm, double *d_frac, int32_t *d_keep,
                           double *d_sum_fpkm, void *stream);

/* TPM = 1e6 * FPKM / total_fpkm (alignments.cpp:1825-1829).  d_total_fpkm[1] is
 * the GLOBAL sum: after the RCCL all-reduce when loci are sharded over ranks.    */
int sbgpu_tpm_device(sbgpu_ctx_t *ctx, int64_t n_iso, const double *d_fpkm,
                     const int32_t *d_keep, const double *d_total_fpkm,
                     double *d_tpm, void *stream);

/* ---- the collective: loci sharded over one process per GPU -------------------------------
 * Replaces the two cross-locus sums of the reference -- `_total_mapped_reads +=`
 * (src/alignments.cpp:1372, an atomic<int> the locus threads add to) and the FPKM total
 * of Sample::procSample (src/alignments.cpp:1821-1824) -- by all-reduce(sum) over the
 * ranks: RCCL over xGMI, a few bytes, once before and once after the EM.  No locus data
 * ever crosses ranks.  Bootstrapping is the caller's (no MPI in the library): rank 0
 * makes an id with sbgpu_comm_unique_id and hands its 128 bytes to the other ranks by
 * whatever channel the driver has (a file, a socket, argv); every rank then calls
 * sbgpu_comm_init(ctx, rank, world, id) -- a collective call -- on the GPU of its ctx.
 * world == 1 needs no id (NULL) and no RCCL: its all-reduce is the identity.  (With
 * SBGPU_COMM_FORCE_RCCL=1 in the environment a world of one takes the RCCL path all the
 * same -- its own id, ncclCommInitRank(1), ncclAllReduce -- so that a one-GPU box can
 * exercise the binding; sbgpu_comm_rccl_ranks then reports 1 instead of 0.)
 * The all-reduces are in place, on device buffers, asynchronous on `stream`.             */
#define SBGPU_COMM_ID_BYTES 128
typedef struct sbgpu_comm sbgpu_comm_t;
int sbgpu_comm_unique_id(uint8_t id_out[SBGPU_COMM_ID_BYTES]);
int sbgpu_comm_init(sbgpu_ctx_t *ctx, int rank, int world, const uint8_t id[SBGPU_COMM_ID_BYTES],
                    sbgpu_comm_t **comm_out);
int sbgpu_comm_info(const sbgpu_comm_t *comm, int *rank, int *world);
/* ranks in the RCCL communicator behind `comm` (ncclCommCount); 0 when the communicator is a
 * world of one that never opened RCCL.                                                     */
int sbgpu_comm_rccl_ranks(const sbgpu_comm_t *comm, int *ranks);
int sbgpu_comm_destroy(sbgpu_comm_t *comm);
int sbgpu_allreduce_sum_f64(sbgpu_comm_t *comm, double *d_buf, int64_t n, void *stream);
int sbgpu_allreduce_sum_i64(sbgpu_comm_t *comm, int64_t *d_buf, int64_t n, void *stream);
/* The same for HOST buffers of up to 512 values (what a driver that keeps its totals on the
 * host calls: `_total_mapped_reads`, the FPKM total): staged through device memory on the
 * context's stream; returns when the result is in `buf`.                                 */
int sbgpu_allreduce_sum_f64_host(sbgpu_comm_t *comm, double *buf, int64_t n);
int sbgpu_allreduce_sum_i64_host(sbgpu_comm_t *comm, int64_t *buf, int64_t n);
/* all-reduce(max): what lets the ranks agree on the LENGTH of an array they are about to sum -- the histogram of the
 * empirical insert-size law (sbgpu_quantify_resident), whose length is the longest locus of any rank's shard.      */
int sbgpu_allreduce_max_i64(sbgpu_comm_t *comm, int64_t *d_buf, int64_t n, void *stream);
int sbgpu_allreduce_max_i64_host(sbgpu_comm_t *comm, int64_t *buf, int64_t n);
/* A communicator whose exchange is the CALLER's (a driver that already has MPI, sockets or -- the tests -- gloo, or whose
 * ranks share a GPU, which RCCL does not serve): `fn(user, buf, n, is_f64, op)` must all-reduce the n 8-byte values of the
 * HOST buffer `buf` in place over the ranks (is_f64: doubles, else int64; op 0 = sum, 1 = max) and return 0.  Every
 * sbgpu_allreduce_* call on such a communicator stages its buffer through host memory and synchronises `stream`.      */
typedef int (*sbgpu_host_allreduce_fn)(void *user, void *buf, int64_t n, int32_t is_f64, int32_t op);
int sbgpu_comm_init_host(sbgpu_ctx_t *ctx, int rank, int world, sbgpu_host_allreduce_fn fn, void *user,
                         sbgpu_comm_t **comm_out);

/* ---- bin-weight model: what fills F ----------------------------
 * Replaces LocusContext::set_theory_bin_weight, src/estimate.cpp:201-234, with
 * ExonBin::effective_len (include/isoform.h:419-516) and InsertSize::emp_dist_pdf
 * (src/read.cpp:274-297).  One work item per (exon bin, isoform) pair:
 *   seg_lens[seg_off[p] .. seg_off[p+1])  lengths of the isoform's segments the bin
 *                                         spans (<= 32), as returned through
 *                                         ExonBin::bin_under_iso (isoform.h:363-411)
 *   implicit_mask[p]                      bit k set: segment k is implicit (mate gap)
 *   iso_len[p]                            exonic length L_j (estimate.hpp:98)
 *   out_index[p]                          element of `F` that receives the weight
 *                                         (NULL: F[p]) -- lets the kernel write
 *                                         straight into an EM batch's F array      */
typedef struct {
   double mean, sd;        /* InsertSize::_mean, _sd                                   */
   int32_t use_emp;        /* InsertSize::_use_emp                                     */
   int32_t start_offset;   /* InsertSize::_start_offset (use_emp only)                 */
   int32_t end_offset;     /* InsertSize::_end_offset                                  */
   int32_t total_reads;    /* InsertSize::_total_reads                                 */
   const double *emp_hist; /* InsertSize::_emp_dist[end_offset-start_offset+1], host   */
   int32_t read_len;       /* ReadTable::read_len_mode(), include/read.hpp:150-160     */
   int32_t long_read;      /* long_read_sample: F = 1/L_j (estimate.cpp:236-247)       */
} sbgpu_insert_t;

/* pdf_out[fl] = InsertSize::emp_dist_pdf(fl) for fl in [0, n) (host arrays).      */
int sbgpu_insert_pdf_table(const sbgpu_insert_t *ins, int32_t n, double *pdf_out);

/* Device-resident form.  d_pdf[pdf_len] must cover every fragment length up to
 * the largest sum of a pair's segment lengths.  Asynchronous on `stream`.         */
int sbgpu_binweight_device(sbgpu_ctx_t *ctx, int64_t n_pairs, const int64_t *d_seg_off,
                           const uint32_t *d_seg_lens, const uint32_t *d_implicit_mask,
                           const int32_t *d_iso_len, const int64_t *d_out_index,
                           const double *d_pdf, int32_t pdf_len, int32_t read_len,
                           int32_t lmin_base, int32_t long_read, double *d_F, void *stream);

/* Host-buffer convenience form: builds the pdf table, uploads, runs, downloads
 * weight_out[p] for every pair (out_index is not used), synchronises.            */
int sbgpu_binweight_host(sbgpu_ctx_t *ctx, int64_t n_pairs, const int64_t *seg_off,
                         const uint32_t *seg_lens, const uint32_t *implicit_mask,
                         const int32_t *iso_len, const sbgpu_insert_t *ins, double *weight_out);

/* ---- exon-bin assignment, integer part --------------------------
 * Replaces the two interval tests LocusContext::assign_exon_bin (src/estimate.cpp:135-198)
 * makes for every (fragment, isoform) of a locus:
 *   Contig::is_compatible(hit, isoform)      src/contig.cpp:547-599
 *   LocusContext::overlap_exons(segs, hit)   src/estimate.cpp:115-131
 * The bookkeeping on their results (LocusContext::set_maps, include/estimate.hpp:29-52:
 * bins keyed by the set of touched segments) is host work in the caller.
 *
 * Annotation, CSR:  locus l owns isoforms iso_off[l]..iso_off[l+1]; isoform i owns exons
 * exon_off[i]..exon_off[i+1] (closed coordinates, sorted: Contig::_genomic_feats' S_MATCH
 * entries; its introns are the gaps between consecutive exons); locus l owns the disjoint
 * exon segments seg_off[l]..seg_off[l+1] (LocusContext::_exon_segs, estimate.hpp:80-91).
 * Hits, CSR: hit h belongs to locus hit_locus[h] and owns features feat_off[h]..feat_off[h+1]
 * exactly as Contig::Contig(const PairedHit&) lays them out (src/contig.cpp:216-267):
 * feat_code 0 = S_MATCH, 1 = S_INTRON, 2 = S_GAP (include/contig.h:26-31), closed
 * feat_left..feat_right, sorted; first and last are S_MATCH.
 * Results, bit words with a fixed stride per hit:
 *   compat[h*compat_words + w] bit b  <=>  is_compatible(hit h, isoform 32*w+b of its locus)
 *   key[h*key_words + w]       bit b  <=>  hit h overlaps segment 32*w+b of its locus
 * compat_words / key_words must cover the widest locus (ceil(n/32)); bits past a locus' own
 * isoforms / segments are 0.  A hit without features gets all-zero words.                 */
typedef struct {
   int64_t n_loci;
   const int64_t *iso_off;     /* [n_loci + 1]                     */
   const int64_t *exon_off;    /* [iso_off[n_loci] + 1]            */
   const uint32_t *exon_left;  /* [exon_off[n_iso]]                */
   const uint32_t *exon_right;
   const int64_t *seg_off;     /* [n_loci + 1]                     */
   const uint32_t *seg_left;   /* [seg_off[n_loci]]                */
   const uint32_t *seg_right;
} sbgpu_annotation_t;

typedef struct {
   int64_t n_hits;
   const int32_t *hit_locus;   /* [n_hits]                         */
   const int64_t *feat_off;    /* [n_hits + 1]                     */
   const uint8_t *feat_code;   /* [feat_off[n_hits]]               */
   const uint32_t *feat_left;
   const uint32_t *feat_right;
} sbgpu_hits_t;

/* Device-resident form: every pointer inside the two structs and d_compat / d_key are device
 * pointers (the structs themselves live on the host).  Asynchronous on `stream`.         */
int sbgpu_exonbin_device(sbgpu_ctx_t *ctx, const sbgpu_annotation_t *annot, const sbgpu_hits_t *hits,
                         int32_t compat_words, int32_t key_words, uint32_t *d_compat,
                         uint32_t *d_key, void *stream);

/* Host-buffer convenience form: validates the CSR arrays (SBGPU_EINVAL / SBGPU_ESHAPE when
 * the word counts do not cover a locus), uploads, runs, downloads, synchronises.         */
int sbgpu_exonbin_host(sbgpu_ctx_t *ctx, const sbgpu_annotation_t *annot, const sbgpu_hits_t *hits,
                       int32_t compat_words, int32_t key_words, uint32_t *compat_out,
                       uint32_t *key_out);

/* ---- exon-bin bookkeeping on the host (the callers either side) --------
 * Plain host code (no GPU work): what the reference does per locus with std::map / std::set
 * between the interval tests above and the inputs of the bin-weight and EM kernels.        */

/* The disjoint exon segments of every locus: sort + unique over the isoforms' exons, then
 * IRanges::disjoint (include/estimate.hpp:80-91, include/interval.hpp:150-191).  Fills
 * seg_off[n_loci+1] and up to `cap` entries of seg_left/seg_right (either may be NULL to
 * size the arrays first); returns the total number of segments, or a negative SBGPU_E*.   */
int64_t sbgpu_segments_host(int64_t n_loci, const int64_t *iso_off, const int64_t *exon_off,
                            const uint32_t *exon_left, const uint32_t *exon_right, int64_t *seg_off,
                            uint32_t *seg_left, uint32_t *seg_right, int64_t cap);

/* The feature list of one fragment from its mates' features (each mate: its CIGAR as
 * MATCH / INTRON features, readhit_2_genomicFeats, src/contig.cpp:12-53), as
 * Contig::Contig(const PairedHit&) builds it (src/contig.cpp:216-267): a GAP feature when the
 * mates are apart, sort + merge_genomicFeats (include/contig.h:111-137) when they touch or
 * overlap; n_left or n_right may be 0 (single read).  The out arrays need room for
 * n_left + n_right + 1 features.  Returns the number of features, 0 when the reference
 * rejects the pair ("paired reads ... are not compatible", estimate.hpp:71-79), negative
 * SBGPU_E* on a bad argument.                                                              */
int sbgpu_hit_features(int n_left, const uint8_t *lcode, const uint32_t *lleft, const uint32_t *lright,
                       int n_right, const uint8_t *rcode, const uint32_t *rleft, const uint32_t *rright,
                       uint8_t *code_out, uint32_t *left_out, uint32_t *right_out);

/* The sample of the empirical insert-size distribution, Sample::fragLenDist's inner loop
 * (src/alignments.cpp:1381-1407): frag_len_out[h] = Contig::exonic_overlaps_len(t, hit.left(),
 * hit.right()) (src/contig.cpp:412-426) when hit h is compatible with exactly ONE transcript t of
 * its locus, else -1.  `compat` is the kernel's result (host copy).  Returns how many hits got a
 * length; InsertSize(frag_lens) (src/read.cpp:238-262) takes them in hit order.              */
int64_t sbgpu_frag_lens_host(const sbgpu_annotation_t *annot, const sbgpu_hits_t *hits,
                             int32_t compat_words, const uint32_t *compat, int32_t *frag_len_out);

/* ---- from aligned read pairs to unique hits (host code) ----------------------
 * HitCluster::collapseAndFilterHits (src/alignments.cpp:656-703) followed by Contig(PairedHit) for
 * every unique hit, for the clusters (loci) of a batch:
 *   - pairs of a locus are ordered by (left end, right end) of the pair (src/read.cpp:917-923); the
 *     reference uses std::sort there, ties keep their input order here;
 *   - a pair one of whose mates spans more than mean + 15.45 sd of the locus' mate spans is skipped
 *     (phi((len - mean) / (5 sd)) > 0.999, :667-682; spans of all mates, population sd, common.h:100-110);
 *   - the others add their raw mass to the cluster's mass (:683-684) and are collapsed with the
 *     previous unique hit when both mates are equal -- same start, same CIGAR (:685-697,
 *     src/read.cpp:196-207,897-910) -- summing the masses in double;
 *   - every unique hit becomes features through sbgpu_hit_features; a rejected one (0 features) is
 *     dropped, its mass stays in the cluster's mass.
 * A mate is its MATCH / INTRON features (readhit_2_genomicFeats); a missing mate has no features.
 * pair_mass[p] is the pair's raw mass: 0.5 / NH per mate of a proper pair, 1 / NH for a singleton
 * (src/read.cpp:49-53,116-123).                                                                  */
typedef struct {
   int64_t n_pairs;
   const int32_t *pair_locus;  /* [n_pairs]                                  */
   const double *pair_mass;    /* [n_pairs]                                  */
   const int64_t *left_off;    /* [n_pairs + 1] into the left mates' features  */
   const uint8_t *left_code;
   const uint32_t *left_left, *left_right;
   const int64_t *right_off;   /* [n_pairs + 1] into the right mates' features */
   const uint8_t *right_code;
   const uint32_t *right_left, *right_right;
} sbgpu_pairs_t;
typedef struct sbgpu_uniq sbgpu_uniq_t;
int sbgpu_collapse_pairs_host(int64_t n_loci, const sbgpu_pairs_t *pairs, sbgpu_uniq_t **out);
void sbgpu_uniq_destroy(sbgpu_uniq_t *u);
/* info: 0 unique hits kept, 1 their features, 2 pairs skipped by the span filter, 3 unique hits
 * rejected by Contig(PairedHit), 4 total mapped reads = sum over loci of (int) cluster mass
 * (src/alignments.cpp:1372).                                                                 */
int sbgpu_uniq_info(const sbgpu_uniq_t *u, int64_t info[8]);
/* The unique hits as an sbgpu_hits_t's arrays (+ hit_mass = (float) collapse mass, cluster_mass
 * [n_loci] in double); NULL = skip.                                                           */
int sbgpu_uniq_export(const sbgpu_uniq_t *u, int32_t *hit_locus, int64_t *feat_off, uint8_t *feat_code,
                      uint32_t *feat_left, uint32_t *feat_right, float *hit_mass, double *cluster_mass);

/* The same on the GPU (csrc/collapse_flat.h): the pairs' arrays of `d_pairs` are DEVICE pointers (pair_locus is
 * not read), grouped by locus as locus_pair_off[n_loci + 1] (host) says; the unique hits stay in HBM in
 * sbgpu_hits_t layout -- sbgpu_uniq_dev_hits hands them to sbgpu_exonbin_device / sbgpu_quantify_device -- and
 * only the per-locus offsets and cluster masses come back.  All clusters of the call at once: one stable device-wide
 * radix sort (two where its key would not fit 64 bits) gives the (left, right) order with ties in input order; the span
 * filter is decided from the spans' exact
 * integer moments (the reference's running sum of squared deviations is taken where a decision could depend on its
 * rounding); equal neighbours collapse, their masses added in double in that order (in any order where every mass
 * of the cluster is a multiple of 2^-20: the sums are then exact); Contig(PairedHit)'s features per unique hit.
 * Up to 2^31 pairs per call and mates of up to 512 features; beyond: SBGPU_EUNSUPPORTED (use
 * sbgpu_collapse_pairs_host).  Synchronises on `stream`.
 * The span filter evaluates phi() with the device's exp(): a pair exactly on the 0.999 boundary could fall on the
 * other side than with the host's libm (not observed).                                                        */
typedef struct sbgpu_uniq_dev sbgpu_uniq_dev_t;
int sbgpu_collapse_pairs_device(sbgpu_ctx_t *ctx, int64_t n_loci, const sbgpu_pairs_t *d_pairs,
                                const int64_t *locus_pair_off, void *stream, sbgpu_uniq_dev_t **out);
void sbgpu_uniq_dev_destroy(sbgpu_uniq_dev_t *u);
/* info as sbgpu_uniq_info */
int sbgpu_uniq_dev_info(const sbgpu_uniq_dev_t *u, int64_t info[8]);
/* The unique hits where they are: *d_hits gets device pointers, *d_hit_mass the (float) collapse masses
 * (device), *locus_hit_off [n_loci + 1] (host): the arguments of sbgpu_quantify_device.  Owned by the handle. */
int sbgpu_uniq_dev_hits(const sbgpu_uniq_dev_t *u, sbgpu_hits_t *d_hits, const float **d_hit_mass,
                        const int64_t **locus_hit_off);
/* Host copies (as sbgpu_uniq_export; NULL = skip). */
int sbgpu_uniq_dev_export(const sbgpu_uniq_dev_t *u, int32_t *hit_locus, int64_t *feat_off, uint8_t *feat_code,
                          uint32_t *feat_left, uint32_t *feat_right, float *hit_mass, double *cluster_mass);

/* ---- cluster streaming: which cluster does an alignment record belong to ---------------
 * Sample::nextClusterRefDemand (src/alignments.cpp:1145-1187, quant mode: the clusters are the annotation's genes, in
 * annotation order = sorted by (reference, left)): ONE forward pass over the position-sorted records.  For cluster k
 * the pass resumes where cluster k - 1 stopped; a record that ends before the cluster begins (or lies on an earlier
 * reference) is skipped for good (hit_lt_cluster, :32-37); a record that begins behind the cluster's end (or on a
 * later reference) ends the cluster and is looked at again for the next one (hit_gt_cluster, :39-49); of the others,
 * those whose transcription strand (XS) is known and differs from the cluster's are dropped, the rest join the
 * cluster (addOpenHit).  A record is therefore offered to exactly one cluster -- the first whose end it does not
 * lie behind -- even where genes overlap.
 *   clusters (HOST arrays): reference id, [left, right] and strand (1 +, 2 -, 0 unknown) of every cluster, in order;
 *   records: reference id, first and last aligned base, flags (bits 2-3: XS strand as in sbgpu_reads_t), sorted by
 *   (reference, left) as the BAM is.
 * Out: read_cluster[i] = the record's cluster or -1, and cluster_read_off[n_clusters + 1] (host): cluster k was offered
 * the records [off[k], off[k + 1]) -- the ones with read_cluster == k among them are its members, in arrival order;
 * with `flags_inout` given, the others get SBGPU_READ_SKIP set, so that range is sbgpu_pair_mates_*'s input as it
 * stands.  Device form: one binary search per cluster, a prefix maximum on the host (clusters are few), one binary
 * search per record.                                                                                            */
typedef struct {
   int64_t n_clusters;
   const int32_t *ref;
   const uint32_t *left, *right;
   const uint8_t *strand;
} sbgpu_clusters_t;
#define SBGPU_READ_SKIP 16u /* (sbgpu_reads_t.flags) the record is not this cluster's: sbgpu_pair_mates_* pass over it */
int sbgpu_assign_reads_host(const sbgpu_clusters_t *clusters, int64_t n_reads, const int32_t *read_ref, const uint32_t *read_left,
                            const uint32_t *read_right, uint8_t *flags_inout, int32_t *read_cluster, int64_t *cluster_read_off);
int sbgpu_assign_reads_device(sbgpu_ctx_t *ctx, const sbgpu_clusters_t *clusters, int64_t n_reads, const int32_t *d_read_ref,
                              const uint32_t *d_read_left, const uint32_t *d_read_right, uint8_t *d_flags_inout,
                              int32_t *d_read_cluster, int64_t *cluster_read_off, void *stream);

/* ---- mate pairing: alignment records -> read pairs --------------------------------------
 * HitCluster::addOpenHit + addHit (src/alignments.cpp:423-655): the records of a cluster, in the order the
 * position-sorted BAM gives them, become PairedHits.
 *   - a record whose span exceeds kMaxFragSpan (1 000 000, src/common.cpp:17) is refused (:512-518);
 *   - a record without a partner (partner_pos 0) or with its partner on another reference is a hit of its own: the
 *     RIGHT mate of a pair without a left one when it is on the reverse strand, else the left mate (:535-545);
 *   - otherwise the cluster's open mates of the same read id are searched, oldest first, for one that starts where
 *     this record's partner starts, expects its partner where this record starts and whose strand (XS) agrees or is
 *     unknown on either side (:590-623): the pair is complete and joins the cluster's hits NOW (the order of the hits
 *     = the order in which pairs are completed); else the record waits as an open mate -- the left one when its
 *     partner lies behind it, the right one when before it; a record whose partner starts where it starts is
 *     refused (:559-585, 630-641);
 *   - what still waits when the cluster is closed is dropped (clearOpenMates, :653).
 * Pair mass: the two reads' masses, 0.5 / NH each; a single read's 1 / NH (src/read.cpp:49-53).
 * The result has the layout of sbgpu_pairs_t -- mates as MATCH / INTRON feature lists (readhit_2_genomicFeats, src/contig.cpp:12-53:
 * an INTRON between two blocks, none where they touch -- an insertion in the read) --:
 * the input of sbgpu_collapse_pairs_host / _device.  One difference a caller should know: the reference's cluster
 * keeps the span of EVERY accepted record for its span filter (:527), the collapse here sees the spans of the
 * mates of the pairs only -- records that never find their mate do not count.                                   */
typedef struct {
   int64_t n_reads;
   const uint64_t *read_id;      /* [n_reads] ReadTable::get_id(read name): the mates of a pair share it      */
   const int64_t *block_off;     /* [n_reads + 1] the record's aligned blocks (the M runs of an M / N CIGAR)  */
   const uint32_t *block_left, *block_right; /* closed coordinates, ascending                                 */
   const uint32_t *partner_pos;  /* [n_reads] where the mate starts; 0: none                                  */
   const uint8_t *flags;         /* [n_reads] bit 0: reverse strand (BAM_FREVERSE); bit 1: the partner lies on
                                    another reference; bits 2-3: transcription strand (XS) 0 unknown, 1 +, 2 - */
   const int32_t *nh;            /* [n_reads] NH tag                                                          */
} sbgpu_reads_t;
#define SBGPU_READ_REVERSE 1u
#define SBGPU_READ_PARTNER_ELSEWHERE 2u
typedef struct sbgpu_matepairs sbgpu_matepairs_t;
/* Host form: `reads` host arrays, the records of cluster l are [locus_read_off[l], locus_read_off[l + 1]). */
int sbgpu_pair_mates_host(int64_t n_loci, const sbgpu_reads_t *reads, const int64_t *locus_read_off, sbgpu_matepairs_t **out);
/* Device form (csrc/matepair_flat.h): `d_reads` device arrays, locus_read_off host.  All clusters of the call at once:
 * one stable device-wide radix sort on a 32-bit key (a group of neighbouring clusters, then a hash of (cluster, read id))
 * brings a read id's records together in arrival order, the first record of every read id walks its group with the
 * reference's open-mate rules (any number of mates of one read id may wait), a compaction of the completing records in
 * arrival order ranks the pairs, and the pairs are written where sbgpu_collapse_pairs_device reads them -- nothing but
 * per-cluster offsets and four counters comes back.
 * Up to 2^31 records and 2^23 clusters per call; beyond: SBGPU_EUNSUPPORTED.                                        */
int sbgpu_pair_mates_device(sbgpu_ctx_t *ctx, int64_t n_loci, const sbgpu_reads_t *d_reads, const int64_t *locus_read_off,
                            void *stream, sbgpu_matepairs_t **out);
void sbgpu_matepairs_destroy(sbgpu_matepairs_t *m);
/* info: 0 pairs (single reads included), 1 complete pairs, 2 single reads, 3 records refused, 4 records that never
 * found their mate, 5 features of the left mates, 6 of the right mates, 7: 0 host arrays; else bit 0 set (the arrays live on
 * the device), bit 1: the positional matching served the call (no sort: the records of every cluster ascend by position and
 * no read id is aligned twice in a cluster), else bits 2.. say why the sorted form did: 1 records not in position order,
 * 2 a read id with more than one fitting mate (0: SBGPU_PAIR_FORCE_SORT).                                                */
int sbgpu_matepairs_info(const sbgpu_matepairs_t *m, int64_t info[8]);
/* The pairs where they are (host or device arrays, see info[7]) + locus_pair_off [n_loci + 1] (host): the arguments
 * of sbgpu_collapse_pairs_host / _device.  pair_locus is set for the host form only.  Owned by the handle.      */
int sbgpu_matepairs_pairs(const sbgpu_matepairs_t *m, sbgpu_pairs_t *pairs, const int64_t **locus_pair_off);
/* Host copies of everything (NULL = skip): pair_mass [pairs], left_off / right_off [pairs + 1], the feature arrays. */
int sbgpu_matepairs_export(const sbgpu_matepairs_t *m, double *pair_mass, int64_t *left_off, uint8_t *left_code,
                           uint32_t *left_left, uint32_t *left_right, int64_t *right_off, uint8_t *right_code,
                           uint32_t *right_left, uint32_t *right_right);

/* ---- BAM alignment records -> the read stream (replaces BAMHitFactory::getHitFromBuf) --------
 * /root/reference/src/read.cpp:480-715: a BAM record becomes a ReadHit -- or is refused -- on its flag word, its CIGAR,
 * the XS / NM / NH tags and four option globals.  In: the UNCOMPRESSED record stream behind the BAM header (what
 * samtools' bam_read1 reads after BGZF inflate, which stays with the caller): per record int32 block_size, the 32-byte
 * core, read name, CIGAR, sequence, qualities, tags.  Out: the accepted records, in file order, as the arrays
 * sbgpu_assign_reads_* and sbgpu_pair_mates_* take, plus why each of the others was refused.
 *   - refused: unmapped (flag 0x4 or no reference); a CIGAR operation of length 0; an operation other than M I D N S H P;
 *     an N outside [min_intron, max_intron]; an I or D that is not between two M, or that is among the first two operations
 *     the reference keeps (H and P are not kept: "10M2I10M" is refused, "3S10M2I10M" is not -- the reference's `i-1 <= 0`,
 *     :594); at most one aligned base; NH > 1 or a secondary alignment while unique_only;
 *   - interval [pos + 1, pos + M + D + N lengths], 1-based closed; blocks = the M runs, a D extends the block before it,
 *     an I leaves two blocks that touch (readhit_2_genomicFeats, src/contig.cpp:12-53: no INTRON between them);
 *   - flags as sbgpu_reads_t's: bit 0 reverse (the record's 0x10), bit 1 partner on another reference (the mate's id differs;
 *     no mate reference included), bits 2-3 the strand: from XS:A ('+' / '-'), else from the library type and the
 *     first-in-pair / reverse bits (:636-651);
 *   - read id = FNV-1 of the read name (ReadTable::get_id, include/read.hpp:164-173); reference id = the file's (the
 *     reference numbers the @SQ lines in order); partner_pos = mate position + 1 (0: none); NH 1 when absent; NM as the
 *     reference keeps it (through an unsigned char); read_len = M + S + I (ReadHit::read_len);
 *   - tags are found as samtools 0.1.19 finds them (external/samtools-0.1.19/bam_aux.c:28-47): a `d` value is stepped over as
 *     if it had no payload; no read leaves the record (a record whose own lengths exceed its block_size: TRUNCATED).  */
typedef struct {
   int32_t min_intron;  /* kMinIntronLength, 20 (src/common.cpp:21; -j)                     */
   int32_t max_intron;  /* kMaxIntronLength, 300000 (src/common.cpp:20; -J)                 */
   int32_t unique_only; /* use_only_unique_hits, 1 (src/common.cpp:67; --allow-multimapped-hits: 0)   */
   int32_t library;     /* 0 unstranded, 1 fr_strand, 2 rf_strand (src/common.cpp:68-69)    */
   int32_t n_ref;       /* references in the header: a larger id is refused; 0: not checked */
} sbgpu_bam_opts_t;
enum {
   SBGPU_BAM_OK = 0,
   SBGPU_BAM_UNMAPPED = 1,
   SBGPU_BAM_BAD_REF = 2,
   SBGPU_BAM_ZERO_OP = 3,
   SBGPU_BAM_OP = 4,
   SBGPU_BAM_INTRON_LONG = 5,
   SBGPU_BAM_INTRON_SHORT = 6,
   SBGPU_BAM_INDEL = 7,
   SBGPU_BAM_SHORT = 8,
   SBGPU_BAM_MULTI = 9,
   SBGPU_BAM_TRUNCATED = 10
};
typedef struct sbgpu_bamreads sbgpu_bamreads_t;
/* The records' offsets in the stream: rec_off[0 .. n] (rec_off[n] = n_bytes).  Returns n, or -1 (sbgpu_last_error) when
 * the stream ends inside a record or `cap` records are not enough.  A caller that inflates BGZF blocks itself can note
 * the offsets as it goes instead.                                                                                     */
int64_t sbgpu_bam_index_host(const uint8_t *bytes, int64_t n_bytes, int64_t *rec_off, int64_t cap);
/* Host form: host arrays in, host arrays in the handle. */
int sbgpu_bam_decode_host(const uint8_t *bytes, int64_t n_bytes, const int64_t *rec_off, int64_t n_records,
                          const sbgpu_bam_opts_t *opts, sbgpu_bamreads_t **out);
/* Device form (csrc/bamdecode_device.h): `d_bytes` and `d_rec_off` device arrays; one lane per record decides and counts
 * its blocks, two scans over the waves' totals place the accepted records and their blocks, a second pass writes them.  The
 * handle's arrays stay on the device: sbgpu_bamreads_reads hands them to sbgpu_assign_reads_device /
 * sbgpu_pair_mates_device as they are.  A record whose offsets do not ascend inside [0, n_bytes] is TRUNCATED, not read.                                                                                */
int sbgpu_bam_decode_device(sbgpu_ctx_t *ctx, const uint8_t *d_bytes, int64_t n_bytes, const int64_t *d_rec_off, int64_t n_records,
                            const sbgpu_bam_opts_t *opts, void *stream, sbgpu_bamreads_t **out);
void sbgpu_bamreads_destroy(sbgpu_bamreads_t *b);
/* info: 0 records, 1 accepted, 2 aligned blocks of the accepted, 3: 1 when some record that got as far as the
 * reference's :605 carries the paired flag (the reference then clears SINGLE_END_EXP), 4: 1 when the arrays live on the
 * device, 5 + s: records of status s (SBGPU_BAM_OK .. SBGPU_BAM_TRUNCATED).                                            */
int sbgpu_bamreads_info(const sbgpu_bamreads_t *b, int64_t info[16]);
/* The accepted records where they are (host or device, info[4]): `reads` for sbgpu_pair_mates_*, and reference id /
 * first / last aligned base for sbgpu_assign_reads_*.  Owned by the handle.                                            */
int sbgpu_bamreads_reads(const sbgpu_bamreads_t *b, sbgpu_reads_t *reads, const int32_t **read_ref, const uint32_t **read_left,
                         const uint32_t **read_right);
/* Host copies (NULL = skip): status [records]; per accepted record its index in the stream, read id, reference, interval,
 * partner_pos, flags, NH, NM, read_len, the record's flag word; block_off [accepted + 1], the blocks.                  */
int sbgpu_bamreads_export(const sbgpu_bamreads_t *b, uint8_t *status, int64_t *record, uint64_t *read_id, int32_t *ref, uint32_t *left,
                          uint32_t *right, uint32_t *partner_pos, uint8_t *flags, int32_t *nh, int32_t *nm, int32_t *read_len,
                          uint32_t *sam_flag, int64_t *block_off, uint32_t *block_left, uint32_t *block_right);

/* LocusContext::assign_exon_bin + set_maps (src/estimate.cpp:135-198, estimate.hpp:29-52)
 * on the kernel's results (host copies of compat / key), hits visited in input order inside
 * each locus (= HitCluster::uniq_hits() order):
 *   - a hit with no compatible isoform is dropped;
 *   - bins are keyed by `key` and numbered in order of first appearance
 *     (UniqPushAndReturnIdx);
 *   - a bin's count is ExonBin::read_count (include/isoform.h:285-296): fragments with the
 *     same (offset, length) feature sequence count once (std::set<Contig>, first one wins),
 *     masses are added in float in that set's order and truncated to int (estimate.cpp:288);
 *   - a bin is paired with every isoform some hit of it is compatible with; for each pair
 *     ExonBin::bin_under_iso (include/isoform.h:363-411) yields the bin-weight kernel's
 *     segment lengths and implicit mask, and pair_out_index its element of the EM batch's F.
 * hit_mass[h] = (float) PairedHit::collapse_mass().  The handle owns the results.          */
typedef struct sbgpu_bins sbgpu_bins_t;
int sbgpu_bins_create(const sbgpu_annotation_t *annot, const sbgpu_hits_t *hits, const float *hit_mass,
                      int32_t compat_words, int32_t key_words, const uint32_t *compat,
                      const uint32_t *key, sbgpu_bins_t **out);
/* The same bins with the per-hit work on the GPU (csrc/bins_device.h): one workgroup per locus
 * groups its hits through a hash table in LDS, ranks the bins by first appearance, accumulates
 * compat unions and masses; only the per-bin arrays come back to the host, where the (bin, isoform)
 * pairs are made as in sbgpu_bins_create.  `annot` holds HOST pointers (the pairs need the
 * annotation on the host); d_hits' pointers, d_hit_mass, d_compat and d_key are DEVICE pointers
 * (the kernel's inputs and results never leave HBM); locus_hit_off[n_loci+1] (host) says where each
 * locus' hits start -- hits must be grouped by locus.  d_hit_bin (device, [n_hits], may be NULL)
 * receives the global bin of every hit; the handle's own hit_bin stays empty.
 * The device form is exact only where the order of the reference's float accumulation cannot
 * matter, and it checks that: hits of a locus sorted by (left end, right end) (HitCluster's order),
 * whole-number masses below 2^24 per bin (the reference's default: no multi-mapped reads), at most
 * 5600 bins per locus.  Otherwise it returns SBGPU_EUNSUPPORTED and the caller uses
 * sbgpu_bins_create.  Synchronises on `stream`.                                                  */
int sbgpu_bins_create_device(sbgpu_ctx_t *ctx, const sbgpu_annotation_t *annot, const sbgpu_hits_t *d_hits,
                             const float *d_hit_mass, const int64_t *locus_hit_off, int32_t compat_words,
                             int32_t key_words, const uint32_t *d_compat, const uint32_t *d_key,
                             int64_t *d_hit_bin, void *stream, sbgpu_bins_t **out);
void sbgpu_bins_destroy(sbgpu_bins_t *bins);
/* info: 0 n_loci, 1 n_iso, 2 n_bins, 3 n_elem (= f_off[n_loci]), 4 n_pairs, 5 total pair
 * segments, 6 hits that landed in a bin, 7 key_words.                                      */
int sbgpu_bins_info(const sbgpu_bins_t *bins, int64_t info[8]);
/* Which grouping made the handle's bins: *on_device = 1 the device kernels (sbgpu_bins_create_device, or the device stage of
 * sbgpu_quantify_host / _device), 0 the library's host code (sbgpu_bins_create).  sbgpu_quantify_host takes the host code
 * when the device form declines (SBGPU_EUNSUPPORTED above); *why_host (optional) then points at the reason, a string owned
 * by the handle ("" when the host code was simply what the caller called).  The results are the same either way; the
 * host path is slower, so a caller that cares can tell.                                                              */
int sbgpu_bins_grouping(const sbgpu_bins_t *bins, int32_t *on_device, const char **why_host);
/* Copies out what the caller asks for (NULL = skip).  The first five arrays are exactly an
 * sbgpu_batch_t minus F (row_off/iso_off/f_off [n_loci+1], count [n_bins]) plus iso_len
 * [n_iso]; bin_key [n_bins*key_words], bin_compat [n_bins*compat_words], hit_bin [n_hits]
 * (global bin of each hit, -1 = dropped); the pair arrays are sbgpu_binweight_device's
 * inputs ([n_pairs+1], [total pair segments], [n_pairs] x3).                               */
int sbgpu_bins_export(const sbgpu_bins_t *bins, int64_t *row_off, int64_t *iso_off, int64_t *f_off,
                      int32_t *count, int32_t *iso_len, uint32_t *bin_key, uint32_t *bin_compat,
                      int64_t *hit_bin, int64_t *pair_seg_off, uint32_t *pair_seg_lens,
                      uint32_t *pair_implicit_mask, int32_t *pair_iso_len, int64_t *pair_out_index);

/* ---- the whole path in one call: fragments -> exon bins -> bin weights -> EM ------------------
 * LocusContext's constructor + the EM of estimate_abundances for every locus of a batch
 * (include/estimate.hpp:60-103, src/estimate.cpp:279-308), host buffers in, host buffers out:
 * uploads the annotation and the hits once, runs sbgpu_exonbin_device, groups the hits into bins
 * on the device (with the library's host code, sbgpu_bins_create, when the device form declines --
 * hits that do not come grouped by locus, ...: same bins, slower; sbgpu_bins_grouping says which ran
 * and why), builds the pairs, runs
 * sbgpu_binweight_device straight into the EM batch's F, plans and runs sbgpu_em_run_device.
 * Nothing but the per-bin arrays, theta and (on request) F comes back over PCIe.
 *   annot          host arrays incl. the segments (sbgpu_segments_host)
 *   hits, hit_mass host arrays; hits grouped by locus and, inside a locus, in uniq_hits() order
 *   insert         the insert-size law (-i); NULL = none given: the empirical distribution is built
 *                  from the hits first (Sample::fragLenDist, src/alignments.cpp:1363-1407,
 *                  Strawberry.cpp:345-355) and written to *insert_used with emp_hist pointing into
 *                  the returned handle
 *   theta_out [n_iso], status_out / iters_out [n_loci]   as sbgpu_em_batch
 *   compat_out [n_hits * ceil(max isoforms / 32)]  optional (NULL): the kernel's compat words
 *   *bins_out      the bins (sbgpu_bins_info / _export, incl. hit_bin; sbgpu_bins_export_weights
 *                  for F); destroy with sbgpu_bins_destroy
 * The caller applies the reference's own epilogue to theta (src/estimate.cpp:310-355).          */
int sbgpu_quantify_host(sbgpu_ctx_t *ctx, const sbgpu_annotation_t *annot, const sbgpu_hits_t *hits,
                        const float *hit_mass, const sbgpu_insert_t *insert, int32_t read_len,
                        int32_t long_read, double *theta_out, int32_t *status_out, int32_t *iters_out,
                        uint32_t *compat_out, sbgpu_insert_t *insert_used, sbgpu_bins_t **bins_out);
/* Keep an annotation resident.  The reference reads its GTF once and streams the reads past it
 * (src/Strawberry.cpp:245-275: one GffReader, loadRefmRNAs; :359: Sample::procSample over every cluster); a driver that calls
 * sbgpu_quantify_host / _device batch after batch against the same annotation pins it once: the arrays are uploaded
 * now, the tables the chain makes of an annotation alone (the isoforms' segment lists, the widest locus) are made
 * now, and every later call on this context that is given an annotation with the SAME counts and array addresses uses
 * them instead of uploading 10+ MB and rebuilding the tables per call.  The caller must not change the arrays while
 * they are pinned (they are not compared).  One annotation per context: pinning another replaces it; unpin (or
 * sbgpu_destroy) releases it.  Results are those of the unpinned call, bit for bit.                                    */
int sbgpu_annotation_pin(sbgpu_ctx_t *ctx, const sbgpu_annotation_t *annot);
int sbgpu_annotation_unpin(sbgpu_ctx_t *ctx);
/* The owner's form of unpin: releases the context's pin only if it is the pin of THESE arrays (addresses and counts; the
 * arrays are not read).  Several objects may share a context and each pin its own annotation, the later pin replacing
 * the earlier: the earlier owner's release must then leave the later pin alone (sbgpu_annotation_unpin would drop it and
 * every later call would quietly go back to uploading the annotation).  *released (may be NULL): 1 if a pin went.      */
int sbgpu_annotation_unpin_matching(sbgpu_ctx_t *ctx, const sbgpu_annotation_t *annot, int32_t *released);
/* The same chain for hits that are in HBM already (a driver that decodes or collapses on the device, or
 * that quantifies the same fragments again): d_hits' arrays and d_hit_mass are DEVICE pointers, grouped by
 * locus as locus_hit_off[n_loci + 1] (host) says and sorted inside a locus like HitCluster's uniq_hits();
 * `annot` holds host pointers; insert == NULL builds the empirical law on the device (see sbgpu_quantify_resident, which
 * also returns it).  The hits take the device grouping
 * (sbgpu_bins_create_device) -- where that declines (fractional masses, ...) the call returns
 * SBGPU_EUNSUPPORTED and the caller uses sbgpu_quantify_host.  theta_out / status_out / iters_out are host
 * arrays; the handle holds the bins (no per-hit bin indices and no weights: sbgpu_bins_export_weights on it
 * fails with SBGPU_EINVAL -- a caller that wants F uses sbgpu_quantify_host).                                       */
int sbgpu_quantify_device(sbgpu_ctx_t *ctx, const sbgpu_annotation_t *annot, const sbgpu_hits_t *d_hits,
                          const float *d_hit_mass, const int64_t *locus_hit_off, const sbgpu_insert_t *insert,
                          int32_t read_len, int32_t long_read, double *theta_out, int32_t *status_out,
                          int32_t *iters_out, sbgpu_bins_t **bins_out);
/* The resident path end to end: sbgpu_quantify_device's chain with pass 1 in front of it and the epilogue behind it, the
 * path's collectives INSIDE the call -- what Sample::preProcess + Sample::procSample compute between the unique hits and
 * the numbers print2gtf prints (src/alignments.cpp:1189-1232, 1736-1834), without theta, F or a per-hit array crossing PCIe:
 *   insert == NULL   the reference's DEFAULT mode (no -i): Sample::fragLenDist (alignments.cpp:1363-1410) on the device -- the
 *                    hits that are compatible with exactly one transcript, Contig::exonic_overlaps_len (contig.cpp:412-426),
 *                    an integer histogram --, all-reduced over `comm`, then InsertSize(frag_lens) (read.cpp:238-262) from
 *                    it: every rank holds the law of the WHOLE sample; returned in *insert_used (emp_hist points into the
 *                    handle).  sbgpu_quantify_host / _device with insert == NULL build their law the same way.
 *   mapped_reads     this rank's part of Sample::total_mapped_reads(): sum over its clusters of (int) weighted_mass()
 *                    (alignments.cpp:1372; sbgpu_uniq_dev_info's info[4]); all-reduced over `comm`
 *   params           the reference's globals; total_mapped_reads and insert_mean are NOT read (the call fills them in
 *                    from the all-reduced total and the law in use)
 *   comm             NULL (or a world of one): no exchange.  Otherwise every rank of the communicator must make this
 *                    call: all-reduce(max) of one int64 and all-reduce(sum) of the histogram (empirical mode only),
 *                    all-reduce(sum) of the mapped-read total, and -- the one collective per step of north_star --
 *                    all-reduce(sum) of the ranks' FPKM totals between abundance_kernel and tpm_kernel
 *                    (estimate.cpp:314-345, alignments.cpp:1821-1829)
 *   out              host arrays to fill (any may be NULL) and, on return, the device arrays of all of them -- the
 *                    context's own memory, valid until its next sbgpu_quantify_* call -- plus the totals
 * Declines like sbgpu_quantify_device (SBGPU_EUNSUPPORTED: the caller uses sbgpu_quantify_host).                        */
typedef struct {
   double *theta, *fpkm, *frac, *tpm; /* in: host [n_iso], or NULL                                    */
   int32_t *keep;                     /* in: host [n_iso], or NULL: 0 erased, 1 kept, 2 kept "NA"    */
   int32_t *status, *iters;           /* in: host [n_loci], or NULL                                   */
   const double *d_theta, *d_fpkm, *d_frac, *d_tpm; /* out: device [n_iso]                            */
   const int32_t *d_keep;                           /* out: device [n_iso]                            */
   const int32_t *d_status, *d_iters;               /* out: device [n_loci]                           */
   int64_t total_mapped_reads;        /* out: Sample::total_mapped_reads(), all ranks                 */
   double total_fpkm;                 /* out: the FPKM total TPM divides by, all ranks                */
   int64_t n_frag_lens;               /* out: size of the empirical sample, all ranks (0: a law was given) */
} sbgpu_abundances_t;
int sbgpu_quantify_resident(sbgpu_ctx_t *ctx, const sbgpu_annotation_t *annot, const sbgpu_hits_t *d_hits,
                            const float *d_hit_mass, const int64_t *locus_hit_off, const sbgpu_insert_t *insert,
                            int32_t read_len, int32_t long_read, int64_t mapped_reads,
                            const sbgpu_abundance_params_t *params, sbgpu_comm_t *comm, sbgpu_insert_t *insert_used,
                            sbgpu_abundances_t *out, sbgpu_bins_t **bins_out);
/* ---- records in HOST memory -> abundances, chunk by chunk, with a bounded footprint on the device -------------------------
 * The reference streams: Sample::nextClusterRefDemand / procSample hold one cluster's reads at a time
 * (src/alignments.cpp:1145-1187, 1736-1811).  The device entries above take a whole sample's records resident; a caller that
 * inflates a BAM file on the host uses this instead (csrc/front_stream_api.hip):
 *   begin   the clusters of quant mode (sorted by (reference, left), as sbgpu_assign_reads_*; cluster k = locus k of the
 *           annotation given to end), the decoder's options, and the most bytes one chunk will hold: two device buffers of
 *           twice that (a chunk + room for the records carried over from the chunk before)
 *   push    the next chunk of the inflated record stream -- WHOLE records, in file order -- from host memory, with the records'
 *           offsets in the chunk (rec_off[0 .. n_records], rec_off[n_records] = n_bytes; NULL: found here with
 *           sbgpu_bam_index_host).  The bytes start their way to the device, and while they travel the chunk pushed BEFORE is
 *           decoded, assigned to the clusters, and the clusters it completes (a record behind their end has been seen) are
 *           paired, collapsed, and their unique hits added to the stream's store on the device; the records of the first
 *           incomplete cluster onward are decoded again with the next chunk.  The caller must leave the bytes AND the offsets
 *           of a push untouched until the NEXT push (or end) returns.  From page-locked memory (hipHostMalloc, cudaHostRegister'ed
 *           buffers a driver inflates into) the upload runs beside the kernels; from pageable memory it is staged by the
 *           runtime before the call computes (correct, no overlap).  A cluster whose records exceed a chunk: SBGPU_ESHAPE.
 *   end     the last chunk, then ONE sbgpu_quantify_resident over the store (arguments as there; mapped_reads is the
 *           stream's own count): the empirical insert-size law is the whole sample's, TPM needs every locus.
 * Results: those of sbgpu_bam_decode_device .. sbgpu_quantify_resident on the whole sample at once, bit for bit.
 * info: 0 records pushed, 1 accepted records consumed, 2 pairs, 3 unique hits, 4 their features, 5 pairs the span filter
 * dropped, 6 mapped reads, 7 chunks, 8 clusters finished, 9 most bytes carried over, 10 records decoded twice, 11 the LEAST free
 * device memory seen since begin (bytes, hipMemGetInfo after every chunk and after the last stage: blocks the library's pool
 * holds idle count as used), 12 bytes per chunk buffer half, 13: 1 once ended, 14 free device memory at begin.               */
typedef struct sbgpu_front_stream sbgpu_front_stream_t;
int sbgpu_front_stream_begin(sbgpu_ctx_t *ctx, const sbgpu_clusters_t *clusters, const sbgpu_bam_opts_t *opts, int64_t chunk_bytes,
                             sbgpu_front_stream_t **out);
int sbgpu_front_stream_push(sbgpu_front_stream_t *fs, const uint8_t *bytes, int64_t n_bytes, const int64_t *rec_off, int64_t n_records);
int sbgpu_front_stream_end(sbgpu_front_stream_t *fs, const sbgpu_annotation_t *annot, const sbgpu_insert_t *insert, int32_t read_len,
                           int32_t long_read, const sbgpu_abundance_params_t *params, sbgpu_comm_t *comm, sbgpu_insert_t *insert_used,
                           sbgpu_abundances_t *out, sbgpu_bins_t **bins_out);
int sbgpu_front_stream_info(const sbgpu_front_stream_t *fs, int64_t info[16]);
/* the store: the unique hits of the clusters finished so far (device arrays, the stream's), their masses, and where every
 * cluster's hits begin (host, [n_clusters + 1]; complete after end)                                                            */
int sbgpu_front_stream_hits(const sbgpu_front_stream_t *fs, sbgpu_hits_t *d_hits, const float **d_hit_mass, const int64_t **locus_hit_off);
void sbgpu_front_stream_destroy(sbgpu_front_stream_t *fs);
/* F of the EM batch a handle from sbgpu_quantify_host holds: F_out[info[3]] (row-major per locus). */
int sbgpu_bins_export_weights(const sbgpu_bins_t *bins, double *F_out);

/* ---- per-bin sequence statistics ---------------------------------------
 * What the reference's "bias" option (-b genome.fa) adds to the `-f` table and nothing else
 * (src/bias.cpp holds no code): for every exon bin, over the bases of its segments concatenated
 * (ExonBin::bin_dnaseq, include/isoform.h:173-182; FaSeqGetter::fetchSeq, src/fasta.cpp:195-200),
 *   gc[b]       Kmer<string>::GCRatio            include/kmer.h:67-76    (C c G g count; exact)
 *   entropy[b]  Kmer<string>::Entropy(seq, 6)    include/kmer.h:46-65    (natural log; bytes other
 *               than ACGTacgt code as A, :106-124; fp64, ~1e-15 relative to the reference)
 *   flags[b]    bit 0..3 = Kmer<string>::HighGCStrech(seq, 20, 0.8), (20, 0.9), (40, 0.8), (40, 0.9)
 *               include/kmer.h:78-88, in the order of src/alignments.cpp:1626-1629         (exact)
 * genome[0] is base `genome_start` (1-based, the coordinates of the segments) of the chromosome,
 * genome_len bytes as they stand in the FASTA (either case, N, ...).  Bin b = segments
 * seg_off[b] .. seg_off[b+1]-1, closed coordinates, in the bin's own (sorted) order.
 * The reference aborts on bins of 40 bases or fewer (live asserts, kmer.h:20,82); here a window
 * that does not fit gives flag 0, fewer than 6 bases give entropy 0, an empty bin gives gc NaN.
 *
 * Device form: all pointers are device pointers; *d_error (caller zeroes it) gets a non-zero
 * value if a segment lies outside the genome window (that bin's outputs are then 0).
 * genome_len must be below 2^32 (SBGPU_ESHAPE otherwise).  The first call on a device uploads two
 * small tables (synchronous); after that the call is asynchronous on `stream`.              */
int sbgpu_binseq_device(sbgpu_ctx_t *ctx, const uint8_t *d_genome, int64_t genome_start, int64_t genome_len,
                        int64_t n_bins, const int64_t *d_seg_off, const uint32_t *d_seg_left,
                        const uint32_t *d_seg_right, double *d_gc, double *d_entropy, uint8_t *d_flags,
                        int32_t *d_error, void *stream);

/* Host-buffer form: validates (SBGPU_EINVAL for a segment outside the window), uploads, runs,
 * downloads, synchronises.                                                                   */
int sbgpu_binseq_host(sbgpu_ctx_t *ctx, const uint8_t *genome, int64_t genome_start, int64_t genome_len,
                      int64_t n_bins, const int64_t *seg_off, const uint32_t *seg_left,
                      const uint32_t *seg_right, double *gc_out, double *entropy_out, uint8_t *flags_out);

/* ---- output formatting, host only -------------------------------
 * The digits Strawberry prints for FPKM / Frac / TPM: std::to_string(double) (= "%f",
 * src/estimate.cpp:335,344 and src/alignments.cpp:1827) copied into a char[12] by
 * Contig::print2gtf (src/contig.cpp:678-700), i.e. the first 11 characters.           */
int sbgpu_format_value(double v, char out[12]);

/* One transcript block exactly as Contig::print2gtf writes it (src/contig.cpp:636-721):
 * the `transcript` line then one `exon` line per exon; source "Strawberry", score 1000,
 * attributes without a space after ';' and ` exon_id "k";` appended on exon lines.
 * keep: 2 ("NA") prints NA for FPKM and Frac (src/estimate.cpp:321,339).  Writes at most
 * cap-1 bytes + NUL into buf and returns the length needed (snprintf convention).       */
int sbgpu_format_gtf_transcript(char *buf, int cap, const char *chrom, char strand, const char *gene_id,
                                const char *transcript_id, const char *ref_gene_id,
                                const char *ref_gene_name, int n_exons, const int32_t *exon_left,
                                const int32_t *exon_right, double fpkm, double frac, double tpm,
                                int32_t keep);

/* One row of the `-f` context table exactly as Sample::printContext writes it
 * (src/alignments.cpp:1549-1639, columns 1-10; the six sequence columns exist only with
 * BIAS_CORRECTION and a genome FASTA: sbgpu_format_context_row_seq below): tab-separated
 *   sample, sample_frag_count (total mapped reads), gene_id, gene_frag_count, transcripts (comma
 *   list), FPKMs (std::to_string each), conditional_probabilities (the bin's weight per isoform,
 *   to_string_with_precision(.,12) = "%.12g", include/common.h:366-372; 0 for an isoform the bin's
 *   last fragment is not compatible with), class_probabilities (Frac, std::to_string),
 *   path_symbol ("[l-r]" per segment of the bin), path_count (unique hits in the bin), newline.
 * Rows of a locus come in std::map order of the bins' coordinate sets (:1552-1563).  snprintf
 * convention like sbgpu_format_gtf_transcript.                                                */
int sbgpu_format_context_row(char *buf, int cap, const char *sample, int32_t sample_frag_count,
                             const char *gene_id, uint32_t gene_frag_count, int n_iso,
                             const char *const *transcript_ids, const double *fpkm,
                             const double *cond_prob, const double *frac, int n_seg,
                             const uint32_t *seg_left, const uint32_t *seg_right, uint32_t path_count);

/* The same row of a run with `-b genome.fa`: six more columns (src/alignments.cpp:1622-1636) --
 * path_gc_content and path_hexmer_entropy (std::to_string, six decimals) and the four stretch flags
 * (std::to_string(bool): 0 / 1), from sbgpu_binseq_*'s outputs for the bin.                   */
int sbgpu_format_context_row_seq(char *buf, int cap, const char *sample, int32_t sample_frag_count,
                                 const char *gene_id, uint32_t gene_frag_count, int n_iso,
                                 const char *const *transcript_ids, const double *fpkm,
                                 const double *cond_prob, const double *frac, int n_seg,
                                 const uint32_t *seg_left, const uint32_t *seg_right, uint32_t path_count,
                                 double gc, double entropy, uint32_t flags);

#ifdef __cplusplus
}
#endif
#endif /* SBGPU_H_ */
