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
 * per-locus EmSolver-shaped adapter lives above this ABI (strawberry_amd/em.py,
 * INTEGRATION.md shows the C++ one).
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

/* ---- per-locus status: the reference's two bools, src/estimate.cpp:305-308 - */
#define SBGPU_EM_OK 0          /* init()==true,  run()==true, converged (estimate.cpp:480 break) */
#define SBGPU_EM_INIT_EMPTY 1  /* init()==false: no row has a weight > 1e-5 (estimate.cpp:391);
                                  theta = theta0; the caller drops the locus (alignments.cpp:1524) */
#define SBGPU_EM_DENOM_ZERO 2  /* run()==false: a row denominator was 0 (estimate.cpp:451-453);
                                  theta = theta0; the caller proceeds (bool ignored, estimate.cpp:308) */
#define SBGPU_EM_MAXITER 3     /* run()==true after all 1000 iterations (estimate.hpp:237)         */

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
int sbgpu_device_count(void);

/* Bind to HIP device `device`; creates the context's streams and workspace.     */
int sbgpu_init(int device, sbgpu_ctx_t **ctx_out);
int sbgpu_finalize(sbgpu_ctx_t *ctx);
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
 *   d_iters[n_loci]           = E-steps started (0 for INIT_EMPTY)
 * Asynchronous on `stream`.                                                     */
int sbgpu_em_run_device(sbgpu_ctx_t *ctx, const sbgpu_plan_t *plan,
                        const int32_t *d_count, const double *d_F,
                        double *d_theta, int32_t *d_status, int32_t *d_iters,
                        void *stream);

/* Device time of the last sbgpu_em_run_device per kernel kind (HIP events recorded
 * on the stream each kind was launched on, all phases of the kind included):
 * ms[6] = {wave half tile, wave base tile, wave double tile, block, tall block,
 * stream}, 0 for kinds not launched.  Synchronises with those events.            */
int sbgpu_em_last_kernel_ms(sbgpu_ctx_t *ctx, float ms[6]);

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
 *   d_sum_fpkm[1]    += sum of FPKM over kept isoforms of this rank (the operand
 *                    of the TPM all-reduce, alignments.cpp:1821-1824); the caller
 *                    zeroes it.                                                  */
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

#ifdef __cplusplus
}
#endif
#endif /* SBGPU_H_ */
