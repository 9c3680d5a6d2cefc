/* dca_hip.h -- C ABI of the MI355X-native DCA compute core (libdca_hip.so).
 *
 * Plain C: opaque handle, raw pointers and sizes, int status codes.  No torch / C++
 * types cross this boundary and nothing throws across it.  Host buffers are
 * caller-owned, C-contiguous.  A context owns one HIP device and one HIP stream;
 * calls on one context are serialised by the caller, different contexts are
 * independent (no globals except the thread-local error string).
 *
 * Each entry names the reference interface it replaces (paths relative to the
 * reference repository root).
 */
#ifndef DCA_HIP_H
#define DCA_HIP_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- status codes */
#define DCA_OK 0
#define DCA_ERR_ARG (-1)        /* invalid argument / call order */
#define DCA_ERR_IO (-2)         /* cannot open MSA file (reference: throws, plmdca_numerics.cpp:743-746) */
#define DCA_ERR_RESIDUE (-3)    /* character outside the reference's table or short line (reference: std::out_of_range, :752) */
#define DCA_ERR_NOMEM (-4)
#define DCA_ERR_HIP (-5)        /* HIP runtime error; text in dca_last_error() */
#define DCA_ERR_NO_DEVICE (-6)  /* no usable gfx950 device: the product never falls back to a CPU path */
#define DCA_ERR_NOT_SPD (-7)    /* correlation matrix not positive definite (reference: numpy LinAlgError, meanfield_dca.py:542-548) */
#define DCA_ERR_STATE (-8)      /* required earlier stage has not been run */

#define DCA_BIOMOLECULE_PROTEIN 1 /* q = 21, plmdca.py:57 */
#define DCA_BIOMOLECULE_RNA 2     /* q = 5 */

#define DCA_F32 32
#define DCA_F64 64

/* carry_mode of dca_plm_configure */
#define DCA_CARRY_EXACT 0   /* mathematically exact pseudolikelihood gradient (opt-in) */
#define DCA_CARRY_CHUNKED 1 /* reference semantics (plmdca_numerics.cpp:492-530 carry-over), chunk-parallel scan with warm-up */
#define DCA_CARRY_SERIAL 2  /* reference semantics, one strictly serial chain per site (slow; checking mode) */

typedef struct dca_ctx dca_ctx;

const char* dca_last_error(void);
int dca_device_count(void);
const char* dca_version(void);
/* Device blocks of 1 MiB and more are kept in a process-wide cache when a context releases them and are handed to
 * later contexts (DCA_POOL_MAX_BYTES, default 64 GiB per process).  This returns them to the driver; result: bytes released. */
size_t dca_release_cached_memory(void);

/* ------------------------------------------------------------------ drop-in FFI
 * Same symbols, signatures and meaning as the reference's ctypes boundary
 * (pydca/plmdca/plmdcaBackend.cpp:151-156 and :204; bound in pydca/plmdca/plmdca.py:79-89).
 * Returns a malloc()'d block of L*q + L(L-1)/2*q*q floats, or NULL on any error
 * (reason in dca_last_error()); release it with freeFieldsAndCouplings (free()).
 * num_threads is accepted and ignored (the work runs on the GPU). */
float* plmdcaBackend(unsigned short biomolecule, unsigned short num_site_states, const char* msa_file,
                     unsigned int seqs_len, float seqid, float lambda_h, float lambda_J,
                     unsigned int max_iteration, unsigned int num_threads, bool verbose);
void freeFieldsAndCouplings(void* h_and_J);

/* ------------------------------------------------------------------ MSA input
 * dca_read_msa: PlmDCA::readSequencesFromFile (plmdca_numerics.cpp:685-767): one
 * sequence per non-empty, non-'>' line, first L characters, 0-based codes with
 * gap = q-1, exact duplicates dropped keeping the first occurrence.
 * Returns the number of unique rows (>=0) or a DCA_ERR_*; raw_count (optional)
 * receives the number of sequence lines read. */
int dca_read_msa(const char* path, int biomolecule, int L, uint8_t* out, int capacity, int* raw_count);
int dca_count_msa_lines(const char* path);
/* The same reader in ONE pass over the file (what plmdcaBackend itself uses): allocates *rows (unique rows x L bytes; release
 * with dca_host_free) and returns the number of unique rows.  The two-call form above opens the file twice, which a pipe
 * does not survive. */
int dca_read_msa_alloc(const char* path, int biomolecule, int L, uint8_t** rows, int* raw_count);

/* The mfDCA path's FASTA reader, pydca/fasta_reader/fasta_reader.py:81-163 (there through Biopython): multi-line
 * records, upper-casing, every character outside the alphabet is the gap state (:138-149), exact duplicates dropped
 * keeping the first occurrence (:153).  Codes are 0-based with gap = q-1 (the reference's Python states minus one).
 * dca_fasta_shape: number of records with residues and their common length (DCA_ERR_ARG if lengths differ).
 * dca_read_fasta: out = capacity x L bytes; returns the number of unique rows; raw_count = records read.  Files with
 * non-ASCII bytes return DCA_ERR_RESIDUE (the Python side then reads them in text mode itself). */
int dca_fasta_shape(const char* path, int* n_records, int* L_out);
int dca_read_fasta(const char* path, int biomolecule, int L, uint8_t* out, int capacity, int* raw_count);
/* The same in ONE pass over the file: the reader allocates *rows (unique rows x *L_out bytes; release with
 * dca_host_free) and returns the number of unique rows. */
int dca_read_fasta_alloc(const char* path, int biomolecule, uint8_t** rows, int* L_out, int* raw_count);
void dca_host_free(void* p);

/* ------------------------------------------------------------------ reference-sequence back-mapping (host)
 * Local pairwise alignment, Smith-Waterman with affine gaps (a gap of length n costs
 * gap_open + (n-1)*gap_extend), standing in for Bio.pairwise2.align.localds as called by
 * SequenceBackmapper.align_pairs_local (sequence_backmapper.py:186-230; biopython 1.74 is a
 * dependency outside the reference tree).  sub: 26 x 26 substitution scores indexed by
 * (letter - 'A').  dca_sw_scores is the search loop of find_matching_seqs_from_alignment
 * (:233-283): best local score of ref against nseq sequences stored back to back in seqs,
 * sequence k = seqs[offsets[k] .. offsets[k+1]).  dca_sw_align returns one optimal alignment:
 * the aligned region (with '-') in aligned_a / aligned_b (capacity la + lb + 1 each) and the
 * 0-based start of the region in each sequence. */
int dca_sw_scores(const char* ref, int lref, const char* seqs, const int* offsets, int nseq, const int* sub,
                  int gap_open, int gap_extend, int* scores_out);
int dca_sw_align(const char* a, int la, const char* b, int lb, const int* sub, int gap_open, int gap_extend,
                 int* score_out, int* start_a, int* start_b, char* aligned_a, char* aligned_b, int* aligned_len);

/* ------------------------------------------------------------------ context */
int dca_create(dca_ctx** out, int device, int precision /* DCA_F32 | DCA_F64 */);
void dca_destroy(dca_ctx* ctx);
/* X: N x L, 0-based codes < q, gap = q-1 (the C++ coding; the Python mfDCA layer
 * converts from the reference's 1-based coding).  Copies to the device; the range of the codes is checked there (a code >= q
 * is DCA_ERR_ARG with the element's position, and the context then holds NO alignment -- not the one it held before). */
int dca_set_msa(dca_ctx* ctx, const uint8_t* X, int N, int L, int q);

/* Sequence weights: PlmDCA::computeSeqsWeight (plmdca_numerics.cpp:611-671) when
 * compare_precision == DCA_F32 ((float)ident/(float)L > (float)seqid) and
 * msa_numerics.compute_sequences_weight (meanfield_dca/msa_numerics.py:13-50) when
 * DCA_F64.  Integer-exact. */
int dca_compute_weights(dca_ctx* ctx, double seqid, int compare_precision);
int dca_set_weights(dca_ctx* ctx, const double* w);           /* externally computed weights */
int dca_get_weights(dca_ctx* ctx, double* w_out);             /* N values: 1/count */
int dca_get_weight_counts(dca_ctx* ctx, uint32_t* counts_out);/* N values (after dca_compute_weights) */
/* Measurement aid (bench.py's roofline of the weights kernel): the integer work the last dca_compute_weights launch really
 * issued -- out[0] = wave x 32-site groups compared (each 16 pairs per lane x (planes xor/or + 1 popcount-add) VALU
 * instructions), out[1] = the same count without the exact early exit, out[2] = bit planes per group (5 or 3). */
int dca_weights_work(dca_ctx* ctx, uint64_t* out3);
int dca_get_meff(dca_ctx* ctx, double* meff_out);

/* ------------------------------------------------------------------ plmDCA
 * dca_plm_configure fixes lambda_h / lambda_J (PlmDCA ctor, plmdca_numerics.cpp:17-48),
 * the carry mode and the scan geometry (chunk sequences per independent scan, warm-up
 * steps; 0 = defaults 128 / 40).  halo = number of leading sequences of this context's
 * MSA that only warm up the scan and do not contribute to fx/g (used by sequence
 * sharding; 0 otherwise).  add_regulariser = 0 on every shard except one. */
int dca_plm_configure(dca_ctx* ctx, double lambda_h, double lambda_J, int carry_mode,
                      int chunk, int warmup, int halo, int add_regulariser);
size_t dca_plm_num_params(int L, int q);
/* Frees the context's plmDCA state (x, g, the N x L q logit / residual tables, optimiser vectors); alignment, weights and
 * communicator stay.  The next dca_plm_configure starts afresh.  For callers that configure a context only to take the
 * initial point from it (the per-rank set-up of a multi-GPU run) -- no counterpart in the reference, whose PlmDCA object owns
 * nothing between plmdcaBackend calls (plmdcaBackend.cpp:151-204). */
int dca_plm_release(dca_ctx* ctx);
/* x <- initial fields/couplings: PlmDCA::initFieldsAndCouplings (plmdca_numerics.cpp:207-249) */
int dca_plm_init_x(dca_ctx* ctx);
int dca_plm_set_x(dca_ctx* ctx, const void* x, int dtype);
int dca_plm_get_x(dca_ctx* ctx, void* x_out, int dtype);
/* fx, g at the context's current x: PlmDCA::gradient (plmdca_numerics.cpp:436-607) */
int dca_plm_gradient(dca_ctx* ctx, double* fx_out);
int dca_plm_get_g(dca_ctx* ctx, void* g_out, int dtype);

/* ------------------------------------------------------------------ native collectives (RCCL over xGMI)
 * One process per GPU; every context owns one RCCL communicator whose collectives are enqueued on the context's own
 * stream, in place on the library's buffers: with it an objective evaluation has no host callback and no device
 * synchronisation around its exchange step.  librccl.so is opened with dlopen: rccl_path if given, else $DCA_RCCL_PATH,
 * else the librccl that lies next to the libamdhip64 this library is bound to (RCCL must run on the same HIP runtime:
 * it receives this library's stream and pointers; a process that also holds PyTorch can contain two HIP runtimes).
 *   rank 0:      dca_comm_unique_id(path, id)           128 bytes; hand them to every rank (store, MPI, file ...)
 *   every rank:  dca_comm_init(ctx, path, id, world, rank)   collective: returns when all ranks have called it
 * No reference counterpart (pydca is single-process); the partition is the one BASELINE.json's north_star names. */
int dca_comm_unique_id(const char* rccl_path, void* id128);
int dca_comm_init(dca_ctx* ctx, const char* rccl_path, const void* id128, int world, int rank);
int dca_comm_destroy(dca_ctx* ctx);
/* For a WATCHDOG thread (the only entry point that may be called while another thread is inside a call on the same context): a
 * peer rank has died; ncclCommAbort releases the communicator and makes the collectives that wait for the peer fail, so the
 * driving thread's call returns an error instead of hanging.  DCA_ERR_STATE if the collective library has no ncclCommAbort. */
int dca_comm_abort(dca_ctx* ctx);
/* world size and rank as the communicator itself reports them (ncclCommCount / ncclCommUserRank): a launcher asserts with
 * it that its N processes form ONE communicator of N ranks.  dca_comm_init / dca_comm_destroy answer DCA_ERR_STATE, and
 * change nothing, while an optimisation whose vectors are cut for the current communicator is in progress. */
int dca_comm_info(dca_ctx* ctx, int* world, int* rank);
/* Small host-side exchange over the context's communicator: every rank hands in n doubles, every rank receives all world * n of
 * them in rank order (timings, status flags of a start-up protocol; collective, synchronises the context's stream). */
int dca_comm_allgather_host(dca_ctx* ctx, const double* mine, int n, double* all);

/* Exchange step of the sharded plmDCA evaluation through the context's communicator (after dca_plm_configure):
 *   mode 1: all-reduce(sum) of the gradient and of fx after every evaluation, optimiser vectors replicated;
 *   mode 2: sharded optimiser vectors -- reduce-scatter(g) per evaluation, all-gather(x) per step, scalar all-reduces
 *           (the scheme of dca_plm_set_vector_sharding; rank / world are the communicator's);
 *   mode 3: mode 2 with the two vector collectives as a DIRECT EXCHANGE -- grouped ncclSend / ncclRecv of slice j
 *           straight to rank j (every xGMI link of the mesh busy at once) and a local sum of the received pieces in
 *           rank order -- for topologies / sizes where RCCL would run its reduce-scatter and all-gather as rings;
 *   mode 0: off.  Replaces any hook set with dca_plm_set_reduce_hook / dca_plm_set_vector_sharding. */
int dca_plm_set_native_comm(dca_ctx* ctx, int mode);

/* Column-strip decomposition of the plmDCA evaluation over the context's communicator (exchange mode 4; instead of
 * dca_plm_configure + dca_plm_set_native_comm).  The context holds the WHOLE alignment and its weights, like a
 * single-GPU one; rank r of `world` holds the columns of sites [L r / world, L (r+1) / world) of the coupling table and of
 * every per-sequence array, walks all sequences for them -- logits, carry scan and scatter stay local, no halo -- and owns
 * the packed parameters of the pairs (i, j), i < j, whose first site it holds (rank 0 the fields too), with their share
 * of the L-BFGS vectors.  Per evaluation two grouped point-to-point exchanges cross the wires: the couplings a rank's
 * columns need from LOWER ranks (and its sites' fields from rank 0) before the table is expanded, and the rows of the
 * gradient table that belong to the sites of lower ranks (and the field gradients for rank 0) before the fold --
 * (L q)^2 / 2 x (1 - 1/world) elements each way summed over the node, against 2 x (world - 1) x P for the sequence-sharded
 * schemes (config D, 8 ranks: 2 x 193 MB against 2 x 1.5 GB), as all-to-all traffic over every xGMI link at once.
 * This is the reference's own parallel axis (its OpenMP loop runs over sites, plmdca_numerics.cpp:490).
 * dca_plm_get_x / _get_g / _scores are collective in this mode (every rank calls them). */
int dca_plm_configure_strips(dca_ctx* ctx, double lambda_h, double lambda_J, int carry_mode, int chunk, int warmup);

/* mfDCA pair counts summed over the shards through the communicator (instead of dca_mf_set_reduce_hook). */
int dca_mf_set_native_comm(dca_ctx* ctx, int on);
/* The decomposition behind MeanFieldDCA(devices = ...): every rank holds the WHOLE alignment and all weights (tens of MB) and
 * counts the sequences [first, first + count) only; with dca_mf_set_native_comm(ctx, 1) the raw pair counts of the windows are
 * summed over the ranks (Meff is the global one everywhere).  Frequencies, correlation matrix, inverse and scores then run as
 * on one GPU on every rank that asks for them; a rank that gives its communicator back afterwards keeps the summed counts.
 * count < 0: the whole alignment again.  (The reference has one process and no counterpart.) */
int dca_mf_set_row_window(dca_ctx* ctx, int first, int count);

/* Sequence weights with the N^2 L / 2 comparisons divided over the ranks (every rank holds the whole alignment --
 * tens of MB -- and counts every world-th tile pair of the upper triangle of the identity matrix; the symmetric
 * half-loop of plmdca_numerics.cpp:646-666, split evenly), then ONE all-reduce(sum) of the N integer counts
 * through the communicator: exact, the same counts on every rank as dca_compute_weights gives. */
int dca_compute_weights_sharded(dca_ctx* ctx, double seqid, int compare_precision);

/* The same split without a communicator: part `part` of `parts` of the comparisons -> partial counts (host array of N,
 * optional); the caller sums the parts by any means and hands the totals back with dca_set_weight_counts. */
int dca_weights_partial_counts(dca_ctx* ctx, double seqid, int compare_precision, int part, int parts, uint32_t* counts_out);
int dca_set_weight_counts(dca_ctx* ctx, const uint32_t* counts);

/* Optional reduction hook for sequence sharding: called after the local data term is
 * on the device, before the optimiser sees it.  g_dev/fx_dev are DEVICE pointers
 * (count elements of dtype / one double); the hook must sum them over all shards in
 * place and return 0.  The stream is idle when the hook runs. */
typedef int (*dca_reduce_hook)(void* user, void* g_dev, size_t count, int dtype, void* fx_dev);
int dca_plm_set_reduce_hook(dca_ctx* ctx, dca_reduce_hook hook, void* user);

/* Sharded optimiser state (optional, on top of sequence sharding).  With a comm hook set the
 * P-vectors of the L-BFGS (x_prev, g, g_prev, d, 5 x (s, y)) are only maintained on this rank's
 * slice: an evaluation ends with REDUCE_SCATTER of the local gradients (instead of the all-reduce of
 * dca_plm_set_reduce_hook, which is then not used), a step ends with ALL_GATHER of x, and the dot
 * products are ALL_REDUCEd as a handful of doubles.  Same bytes on the wire as one all-reduce per
 * evaluation, but the vector work is divided by `world`.  The hook works on DEVICE memory, in place:
 *   DCA_COMM_ALL_REDUCE     buf[0..count) summed over the ranks (dtype DCA_F64 scalars)
 *   DCA_COMM_REDUCE_SCATTER buf[0..count) summed; the rank needs only its slice afterwards
 *   DCA_COMM_ALL_GATHER     the rank's slice of buf is valid; all of buf[0..count) afterwards
 * count is a multiple of world; rank r's slice is [r * count / world, (r + 1) * count / world).
 * The context's stream is idle when the hook runs.  Call after dca_plm_configure. */
enum { DCA_COMM_ALL_REDUCE = 0, DCA_COMM_REDUCE_SCATTER = 1, DCA_COMM_ALL_GATHER = 2 };
typedef int (*dca_comm_hook)(void* user, int op, void* buf_dev, size_t count, int dtype);
int dca_plm_set_vector_sharding(dca_ctx* ctx, int rank, int world, dca_comm_hook hook, void* user);

typedef struct {
    int status;        /* libLBFGS code as the reference would report it (lbfgs.h:76-149): 0, -997, -998, -1001 ... */
    int iterations;    /* completed iterations since dca_plm_lbfgs_begin */
    int evaluations;   /* objective/gradient evaluations since dca_plm_lbfgs_begin */
    int finished;      /* 1 once a terminal status was reached */
    double fx, xnorm, gnorm, step;
    double seconds;    /* wall time spent inside dca_plm_lbfgs_iterate so far */
} dca_plm_stats;

/* L-BFGS with More-Thuente line search and the backend's parameters
 * (plmdcaBackend.cpp:68-75: m=5, epsilon=1e-3, max_linesearch=5, ftol=1e-4; the rest
 * libLBFGS defaults, lbfgs.cpp:116-121).  begin() evaluates at the current x;
 * iterate() runs up to `iterations` more iterations (resumable); max_iterations is the
 * reference's cap (-997 when exceeded; 0 = unlimited). */
int dca_plm_lbfgs_begin(dca_ctx* ctx, int max_iterations, int verbose);
int dca_plm_lbfgs_iterate(dca_ctx* ctx, int iterations, dca_plm_stats* stats_out);
/* Abandons the optimisation in progress (x and g keep their current values): the exchange scheme, the decomposition and the
 * communicator may change again.  A run that reached its cap or converged has ended by itself.  No reference counterpart
 * (lbfgs() runs to completion, lbfgs.cpp:248-644); needed because dca_plm_lbfgs_iterate is resumable. */
int dca_plm_lbfgs_end(dca_ctx* ctx);

/* ------------------------------------------------------------------ one call
 * The reference's single entry plmdcaBackend(biomolecule, num_site_states, msa_file, seqs_len, seqid, lambda_h, lambda_J,
 * max_iteration, num_threads, verbose) (plmdcaBackend.cpp:151-201) with a device list in place of num_threads, either a file (one
 * sequence per line, the reference's reader and its first-occurrence dedup) or a pre-encoded alignment, and what the reference
 * drops -- status, iterations, evaluations, fx, norms, seconds -- in dca_plm_stats.  Several devices: one rank per device as one
 * host thread each (no helper process), column-strip decomposition over the library's RCCL communicators; float64 gives the
 * single-device bytes.  x_out: P = L q + L (L - 1) / 2 q^2 elements of x_dtype (DCA_F32 | DCA_F64), packed as
 * plmdca_numerics.cpp:467-480.  Errors as everywhere: DCA_ERR_* and dca_last_error(). */
typedef struct dca_plm_args {
    int biomolecule;             /* DCA_BIOMOLECULE_PROTEIN (q = 21) | DCA_BIOMOLECULE_RNA (q = 5) */
    const char* msa_file;        /* one sequence per line (as plmdcaBackend reads it), or NULL */
    const uint8_t* msa;          /* ... or num_seqs x seqs_len codes < q, gap = q - 1 (as dca_set_msa takes them) */
    int num_seqs, seqs_len;      /* seqs_len is needed with a file as well (as plmdcaBackend's seqs_len) */
    float seqid, lambda_h, lambda_J;
    int max_iterations;          /* the reference's cap (0: unlimited) */
    int precision;               /* DCA_F32: the reference's arithmetic; DCA_F64: the parity mode */
    const int* devices;          /* GPU indices, one rank each; NULL / num_devices 0: device 0 */
    int num_devices;
    int exchange_scheme;         /* 0 (default) or 4: column strips */
    const char* rccl_path;       /* NULL: the librccl next to the HIP runtime in use (DCA_RCCL_PATH overrides) */
    int verbose;                 /* the reference's per-iteration lines on stderr (rank 0) */
} dca_plm_args;
int dca_plm_run(const dca_plm_args* args, void* x_out, int x_dtype, dca_plm_stats* stats_out);

/* Frobenius-norm scores of the current x: PlmDCA.get_couplings_no_gap_state +
 * compute_sorted_FN / compute_sorted_FN_APC (plmdca.py:246-268, :437-524), in pair
 * order (0,1),(0,2)...; sorting is left to the host. */
int dca_plm_scores(dca_ctx* ctx, int apc, double* scores_out);

/* Direct-information scores of the current x (PlmDCA.compute_direct_info_unsorted_DI /
 * compute_sorted_DI[_APC], plmdca.py:683-790, numerics plmdca/msa_numerics.py:156-311) in pair
 * order.  reg_fi: L*q regularised single-site frequencies (the reference computes them from
 * its Python reader's alignment with pseudocount 0.5, plmdca.py:622-648). */
int dca_plm_di_scores(dca_ctx* ctx, const double* reg_fi, int apc, double* scores_out);

/* (q-1)x(q-1) coupling blocks of the current x for `npairs` site pairs (pairs[2k] < pairs[2k+1]),
 * row-major a,b, as doubles; shift != 0 applies the zero-sum gauge of PlmDCA.shift_couplings
 * (plmdca.py:320-342).  Feeds PlmDCA.compute_params (plmdca.py:345-434). */
int dca_plm_pair_couplings(dca_ctx* ctx, const int* pairs, int npairs, int shift, double* out);

/* ------------------------------------------------------------------ DI on caller-provided arrays
 * The module-level functions of the reference: compute_two_site_model_fields + compute_direct_info
 * (meanfield_dca/msa_numerics.py:378-533: layout 1 = couplings as the n x n matrix, n = L(q-1);
 * plmdca/msa_numerics.py:156-311: layout 2 = gap-stripped blocks, pair order, (q-1)^2 each).
 * All arrays are HOST doubles; reg_fi is L x q; fields_out (optional) receives [pairs][2][q],
 * di_out (optional) [pairs].  No alignment needs to be set on the context. */
int dca_di_from_arrays(dca_ctx* ctx, const double* couplings, int layout, const double* reg_fi, int L, int q,
                       double* fields_out, double* di_out);

/* compute_direct_info with the caller's own two-site model fields (its `fields_ij` argument,
 * meanfield_dca/msa_numerics.py:473-533, plmdca/msa_numerics.py:249-311): fields_ij is [pairs][2][q] HOST doubles
 * and is used as given -- no fixed-point iteration runs.  di_out: [pairs]. */
int dca_di_from_fields(dca_ctx* ctx, const double* couplings, int layout, const double* reg_fi, const double* fields_ij,
                       int L, int q, double* di_out);

/* ------------------------------------------------------------------ ranking
 * Pair indices of the most recent score vector computed on this context (dca_plm_scores,
 * dca_plm_di_scores, dca_mf_scores, dca_mf_di_scores, dca_mf_run) in descending score order,
 * equal scores in ascending pair order: the sorted(..., reverse=True) step of
 * compute_sorted_FN / _APC / DI (meanfield_dca.py:941, plmdca.py:479), done on the device copy. */
int dca_scores_order(dca_ctx* ctx, int32_t* order_out, int capacity);

/* ------------------------------------------------------------------ mfDCA
 * Stage functions mirror pydca/meanfield_dca/msa_numerics.py; all float64. */
int dca_mf_single_site_freqs(dca_ctx* ctx, double* fi_out /* L*q, gap last (:53-89) */);
int dca_mf_pair_site_freqs(dca_ctx* ctx, double* fij_out /* pairs*(q-1)^2 (:182-229) */);
/* regularise (:92-125, :231-267) + build the L(q-1) x L(q-1) correlation matrix (:270-318) */
int dca_mf_corr_mat(dca_ctx* ctx, double pseudocount, double* corr_out /* may be NULL */);
/* couplings = -inv(C) (:321-342) by blocked Cholesky on f64 MFMA */
int dca_mf_couplings(dca_ctx* ctx, double* couplings_out /* may be NULL */);
/* FN / FN_APC of the couplings (meanfield_dca.py:902-988), pair order */
int dca_mf_scores(dca_ctx* ctx, int apc, double* scores_out);
/* DI / DI_APC of the couplings (meanfield_dca.py:793-899; msa_numerics.py:378-533), pair order;
 * needs dca_mf_corr_mat + dca_mf_couplings (or dca_mf_run) first */
int dca_mf_di_scores(dca_ctx* ctx, int apc, double* scores_out);
/* local fields of the global model, L*(q-1) doubles (MeanFieldDCA.compute_fields,
 * meanfield_dca.py:588-633); needs the couplings */
int dca_mf_fields(dca_ctx* ctx, double* fields_out);
/* coupling blocks of selected pairs, optionally gauge shifted (MeanFieldDCA.compute_params,
 * meanfield_dca.py:661-752; shift_couplings :636-658) */
int dca_mf_pair_couplings(dca_ctx* ctx, const int* pairs, int npairs, int shift, double* out);
/* Sequence sharding of the pair counts: with a hook set, every shard context (its block of the
 * alignment, the GLOBAL weights of those sequences via dca_set_weights) calls it once after its
 * local counts are on the device: g_dev = the Lq x Lq raw weighted counts (doubles), fx_dev = Meff
 * (one double); the hook sums both over the shards in place.  Everything downstream (frequencies,
 * correlation matrix, inverse, scores) is then identical on every shard. */
int dca_mf_set_reduce_hook(dca_ctx* ctx, dca_reduce_hook hook, void* user);
/* whole chain on the device: counts -> C -> -inv -> scores */
int dca_mf_run(dca_ctx* ctx, double pseudocount, int apc, double* scores_out, double* couplings_out /* may be NULL */);
/* stage API on caller-provided arrays: construct_corr_mat (:270-318) from regularised
 * frequencies (reg_fi: L*q, reg_fij: pairs*(q-1)^2) -> corr_out: (L(q-1))^2 */
int dca_mf_corr_from_freqs(dca_ctx* ctx, const double* reg_fi, const double* reg_fij, int L, int q, double* corr_out);
/* test hook / stage API: inverse of a host SPD matrix through the same device path */
int dca_spd_inverse(dca_ctx* ctx, const double* A, int n, double* Ainv_out);

/* ------------------------------------------------------------------ timing
 * When profiling is on, selected kernels are bracketed with HIP events on the
 * context's stream.  dca_get_kernel_time returns accumulated ms and launch count
 * for a kernel tag ("weights", "plm_logits", "plm_softmax", "plm_scatter", "plm_expand",
 * "plm_fold", "lbfgs_vec", "mf_counts", "mf_inverse", "scores"). */
int dca_set_profiling(dca_ctx* ctx, int on);
/* Only the stage of this name ("plm_scatter", "plm_logits", "mf_inverse", ...) is bracketed -- two event records per launch of it
 * instead of two per stage (an event record costs the stream ~5 us: 14 per plmDCA iteration are 5 % of config C's step, 0.4 % of
 * D's).  NULL or "" switches profiling off.  Measurement aid of bench.py's timed region; no reference counterpart. */
int dca_set_profiling_only(dca_ctx* ctx, const char* stage);
int dca_get_kernel_time(dca_ctx* ctx, const char* tag, double* ms_out, int* launches_out);
int dca_reset_kernel_times(dca_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* DCA_HIP_H */
