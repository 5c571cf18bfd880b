/*
 * phyloformer_amd — C ABI of the MI355X (gfx950) Phyloformer inference path.
 *
 * One shared library (libphyloformer_amd.so, built by hipcc for gfx950) owns
 * device memory, the HIP stream, the pre-split weights and every kernel.  The
 * Python host code (phyloformer_amd/engine.py) binds these symbols with
 * ctypes; nothing in the signatures is a torch / numpy type.
 *
 * What each entry point replaces in the reference (lucanest/Phyloformer; the
 * reference has no FFI layer of its own, its boundary is Python-level —
 * SURVEY.md §8b):
 *
 *   pf_create            Phyloformer(**hp) + load_state_dict + .to(device) + .eval()
 *                        infer_alns.py:71-86, phyloformer/model.py:109-164
 *   pf_forward           model(aln[None, :].float())
 *                        infer_alns.py:112 -> phyloformer/model.py:166-187
 *                        (embedding :173, pair expansion :175, 6 x PhyloformerLayer
 *                        :87-106 with ScaledLinearAttention attention.py:160-197,
 *                        pwFNN + Softplus :182, site mean :185)
 *   pf_forward_sharded   the same forward for a contiguous block of sites of every
 *                        pair; row-attention statistics (attention.py:183-190 with
 *                        dim=-2 = sites, model.py:91) and the final site sums
 *                        (model.py:185) are all-reduced over RCCL.  The reference has
 *                        no multi-device inference; this is the build's site-sharding.
 *   pf_destroy           garbage collection of the nn.Module
 *
 * Conventions
 *   - every function returns PF_OK (0) or a negative pf_status; nothing throws
 *     across the ABI; pf_last_error() gives the message for the last failure on
 *     that handle (or, with NULL, the last failure of pf_create on this thread);
 *   - the caller owns all host buffers; the library owns all device memory and
 *     copies the weights at pf_create;
 *   - a handle is bound to one device and one stream and is not thread-safe;
 *     distinct handles are independent;
 *   - residues are alphabet indices 0..21 in the order "ARNDCQEGHILKMFPSTWYVX-"
 *     (phyloformer/data.py:7); the one-hot tensor of the reference never exists;
 *   - pairs are enumerated (i, j), i < j, lexicographically (model.py:13-17),
 *     P = N(N-1)/2 outputs per alignment.
 */
#ifndef PHYLOFORMER_AMD_H
#define PHYLOFORMER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PF_ABI_VERSION 5

typedef enum pf_status {
    PF_OK = 0,
    PF_EINVAL = -1,  /* bad dims, residue index > 21, N > max_seqs, unsupported architecture */
    PF_EHIP = -2,    /* a HIP runtime call or kernel launch failed */
    PF_ERCCL = -3,   /* RCCL missing or a collective failed */
    PF_ENOMEM = -4,  /* device or host allocation failed */
    PF_ESTATE = -5,  /* call not valid in the handle's current state */
    PF_EIO = -6      /* a file could not be read (pf_fasta_batch_load: detail = errno) */
} pf_status;

/* Flat fp32 weight blob.  Order (phyloformer_amd/weights.py::blob_layout):
 *   embedding_block.0.weight [E][22], .bias [E];
 *   per block b, for a in (row, col):
 *       a_norm.weight [E], a_norm.bias [E],
 *       q_proj.weight [H][E], q_proj.bias [H], k_proj.weight [H][E], k_proj.bias [H],
 *       v_proj.weight [E][E], v_proj.bias [E], out_proj.weight [E][E], out_proj.bias [E];
 *     then ffn_norm.weight [E], ffn_norm.bias [E],
 *       ffn.0.weight [4E][E], ffn.0.bias [4E], ffn.3.weight [E][4E], ffn.3.bias [E];
 *   pwFNN.0.weight [E], pwFNN.0.bias [1].
 * The kernels are specialised for E = 64, H = 4 (all shipped checkpoints);
 * other values are refused with PF_EINVAL. */
typedef struct pf_weights_t {
    int32_t n_blocks;
    int32_t n_heads;
    int32_t embed_dim;
    int32_t n_alphabet;     /* 22 */
    const float* blob;
    uint64_t blob_len;      /* number of floats */
} pf_weights_t;

typedef struct pf_handle pf_handle_t;

/* Expected blob_len for an architecture, so bindings can check before calling. */
uint64_t pf_blob_len(int32_t n_blocks, int32_t n_heads, int32_t embed_dim);

int pf_abi_version(void);

/* What this library was built from (ABI 4): a JSON object, static storage -
 *   {"abi", "arch", "hipcc" (HIP and clang versions), "sched_strategy" ("iterative-ilp" | "default"),
 *    "sched_fallback" (true when hipcc could not compile pf_lib.hip with the intended strategy and
 *    phyloformer_amd/build.py was allowed to fall back: 1-7 % slower kernels), "flags" per translation unit,
 *    "source_hash" (sha256/16 over csrc/ + include/), "kernel_hash" (pf_device.hip.h + the flags of its unit)}.
 * bench.py copies it into its line and refuses PMC traffic figures taken with another kernel_hash. */
const char* pf_build_info(void);

/* Create a handle on HIP device `device`: uploads the weights, builds the
 * embedding table and the fp16 hi/lo MFMA operand images.  Fails with PF_EHIP
 * if no gfx950 device is present: there is no CPU fallback. */
int pf_create(const pf_weights_t* w, int device, pf_handle_t** out);
int pf_destroy(pf_handle_t* h);
const char* pf_last_error(const pf_handle_t* h);

/* Options (before or between forwards):
 *   "max_seqs"   int   sequence cap, default 200 (model.py:39); 0 lifts it
 *   "profile"    int   1 = bracket every launch with HIP events (see pf_profile_*); 2 = only the
 *                      dominant kernel k_main (event pairs serialise the stream: 1 costs ~15 %, 2 ~5 %)
 *   "debug_keep" int   1 = keep per-layer activations for pf_debug_read
 *   "force_rccl" int   1 = pf_comm_init creates a real RCCL communicator even for one rank (tests)
 *   "ws_limit_mb" int  workspace budget per batch chunk (default 24576)
 *   "colstats_fine" int k_colstats blocks per pair group (0), per run of a group (1) or chosen from the batch
 *                      size (-1, default): the same summation tree either way, so the same bits; tests/tools
 *   "two_streams" int  0 = a batch runs on one stream; default 1: forwards of >= 2 alignments run as two
 *                      independent half-batches on two streams (same results bit for bit, a few % faster);
 *                      environment PF_TWO_STREAMS=0/1 sets the initial value (A/B runs of whole programs)
 *   "overlap"    int   0 = site-sharded forwards issue one collective per block for the whole batch instead of
 *                      two half-batches on two streams
 *   "reserve_cus" int  CUs the persistent kernels leave to the RCCL kernels while collectives run (default 8)
 *   "materialize_x0" int 1 = k_embed writes x0 = T[a_i] + T[a_j] to HBM and block 0 reads it (round-1 path);
 *                      default 0: block 0's kernels form it from the embedding table on the fly
 *   "embed_mfma" int   1 = compute block 0's row statistics with the MFMA kernel (k_main<FIRST>) instead of
 *                      the residue-pair table lookup (k_embed); cross-check only, same results to fp32 noise
 *   "precise"    int   which alignments take the float64 path (csrc/pf_precise.hip.h): -1 (default) = chosen from
 *                      the alignment's shape (fewer than 32 sites or 8,192 pair-site tokens: the distance is a mean over sites, and on a
 *                      handful of them the fp32 reference itself is 3e-5 ... 8e-4 from its float64 evaluation, so no
 *                      fp32-level kernel can promise 1e-4 against it), 0 = never, 1 = always (any shape in float64;
 *                      3-9 x slower).  The choice never depends on the batch.  Above the option: a checkpoint or
 *                      shape whose operands could overflow the default kernels' fp16 MFMA operands (weights ~ 1000 x
 *                      the trained ones, more than 2^20 sites) always runs in float64.
 *   "recheck_above" int  range re-check of pf_forward / pf_forward_sharded (the host-buffer entry points; only while
 *                      "precise" is -1): an alignment whose largest predicted distance exceeds this many substitutions
 *                      per site (default 8; 0 = off), or is not finite, is computed again on the float64 kernels before
 *                      the call returns.  An absolute tolerance of 1e-4 on a value of 10 asks for 1e-5 relative - the
 *                      rounding level of fp32 arithmetic itself, the reference's included; alignments stay far below
 *                      (<= 5 on the reference's test data), uniformly random residues do not (9-13).  Per alignment,
 *                      never a function of the batch; pf_profile_get("rechecked") counts them.  The device entry
 *                      points do not re-check (their results never pass through the host).
 * Test / tool switches, not part of the contract:
 *   "colstats_ring" int  0 = k_colstats prefetches its rows through registers instead of the per-wave LDS ring (the
 *                      same bits; A/B and counter runs)
 *   "precise_ffn_valu" int 1 = the float64 FFN on the plain VALU kernel instead of the fp64 matrix cores (cross-check)
 *   "phase_prof"  int  1 = in-kernel phase timers of k_main (pf_debug_read "phase_prof")
 *   "ablate"      int  energy experiments: phases of k_main switched off - RESULTS INVALID
 */
int pf_set_option(pf_handle_t* h, const char* key, int64_t value);

/* Forward pass.  idx: host uint8 [B][N][L]; out: host float [B][P].
 * Synchronous: returns after `out` is filled (and re-checked, option "recheck_above").  Never communicates: on a handle that carries a
 * communicator (pf_comm_init) pf_forward / pf_forward_device still process this rank's own
 * alignments only (alignment-level data parallelism); collectives belong to pf_forward_sharded*.
 * Errors: PF_EINVAL for B < 1, N < 2, L < 1, N > max_seqs, or an index > 21. */
int pf_forward(pf_handle_t* h, const uint8_t* idx, int32_t B, int32_t N, int32_t L, float* out);

/* Same with device-resident buffers, asynchronous on the handle's stream
 * (used by the benchmark so the timed region starts with inputs in HBM).
 * d_idx: device uint8 [B][N][L]; d_out: device float [B][P].
 * Residues are NOT validated on this path (the bytes never pass through the host).  A byte > 21 cannot
 * fault: every table lookup of the kernels clamps it to 21 ('-').  It is reported late: the embedding
 * kernel raises a sticky flag on the handle, and the next pf_synchronize / pf_memcpy_d2h on it returns
 * PF_EINVAL once (the results of the forwards since the previous synchronisation are then those of the
 * clamped alignment, not of a valid one).  pf_forward / pf_forward_sharded keep refusing such input up
 * front (the reference raises KeyError, phyloformer/data.py:25-26). */
int pf_forward_device(pf_handle_t* h, const uint8_t* d_idx, int32_t B, int32_t N, int32_t L,
                      float* d_out);

/* Site-sharded forward: this rank holds sites [l_begin, l_end) of an alignment
 * with L_total sites.  idx: host uint8 [B][N][l_end - l_begin].  Every rank
 * receives the full result in out [B][P].  The row-attention statistics are all-reduced once per
 * block and the site sums once at the end (n_blocks + 1 collectives; with B >= 2 and "overlap" = 1 the
 * batch runs as two half-batches on two streams with one communicator each: 2 (n_blocks + 1) collectives).
 * Requires pf_comm_init when the communicator has more than one rank; a partial site range
 * (l_end - l_begin < L_total) on a handle without a communicator fails with PF_ESTATE instead of
 * returning partial sums. */
int pf_forward_sharded(pf_handle_t* h, const uint8_t* idx, int32_t B, int32_t N,
                       int32_t l_begin, int32_t l_end, int32_t L_total, float* out);
int pf_forward_sharded_device(pf_handle_t* h, const uint8_t* d_idx, int32_t B, int32_t N,
                              int32_t l_begin, int32_t l_end, int32_t L_total, float* d_out);
/* A rank without sites (L_total < world size: l_begin == l_end, idx may be NULL) still calls
 * pf_forward_sharded*: it joins every collective of its peers with zeros, cut into the same chunks and
 * halves.  On a handle that does not communicate an empty range is PF_EINVAL like any L < 1. */

/* RCCL bootstrap (one process per GPU).  Rank 0 calls pf_comm_unique_id and
 * ships the PF_UNIQUE_ID_BYTES bytes to the other ranks by any means
 * (torch.distributed store, file, socket); every rank then calls pf_comm_init.
 * The blob is opaque: two ncclUniqueIds, because pf_comm_init creates TWO communicators, one per
 * stream of the handle - a site-sharded forward runs its two half-batches on two streams, and a
 * communicator of its own per stream means RCCL never orders one half's all-reduce behind the
 * other's.  pf_comm_init is collective (every rank, same order) and all-or-nothing: on failure the
 * handle keeps no communicator and stays a working single-rank engine. */
#define PF_UNIQUE_ID_BYTES 256
int pf_comm_unique_id(void* id_out);
int pf_comm_init(pf_handle_t* h, const void* unique_id, int32_t rank, int32_t world_size);
int pf_comm_destroy(pf_handle_t* h);
/* Which librccl the library resolved ($PF_RCCL_LIB, else $ROCM_PATH/lib, /opt/rocm/lib, then the loader's
 * search path) and its ncclGetVersion code.  Fails with PF_ERCCL when no librccl bound to the same HIP
 * runtime as this library can be loaded. */
int pf_comm_info(char* path_out, size_t path_cap, int32_t* version);

/* Stream / device access for callers that time with HIP events. */
int pf_synchronize(pf_handle_t* h);
int pf_get_stream(pf_handle_t* h, void** hip_stream_out);
int pf_device_malloc(pf_handle_t* h, size_t bytes, void** out);
int pf_device_free(pf_handle_t* h, void* p);
int pf_memcpy_h2d(pf_handle_t* h, void* dst, const void* src, size_t bytes);
int pf_memcpy_d2h(pf_handle_t* h, void* dst, const void* src, size_t bytes);

/* Per-kernel HIP-event timing ("profile" = 1).  Names: "embed", "rowfin",
 * "colstats", "colfin", "main", "allreduce", "mha_qkv", "mha_attn", "mha_out".  Totals accumulate
 * until reset.  "collectives" returns the number of all-reduces issued since the last reset in
 * *launches (counted always, no profiling option needed; *total_ms = 0); "rechecked" likewise the number of
 * alignments the range re-check (option "recheck_above") computed again on the float64 kernels. */
int pf_profile_reset(pf_handle_t* h);
int pf_profile_get(pf_handle_t* h, const char* kernel, int64_t* launches, double* total_ms);

/* Debug taps ("debug_keep" = 1), valid after a forward:
 *   "x<k>"    float [B][P][Lloc][64]  residual stream after k main kernels
 *             (x0 = embedding + pair expansion, x<k> = output of block k-1)
 *   "srow<k>" float [B][P][72]   row statistics feeding block k
 *   "ctx<k>"  float [B][Lloc][64] column context of block k
 * (the float64 path keeps only "x<k>", k >= 1, narrowed to float)
 * Returns the number of floats written (<= cap) or a negative status. */
int64_t pf_debug_read(pf_handle_t* h, const char* name, float* dst, int64_t cap);

/* Device properties the benchmark prints: name, CU count, HBM bytes. */
int pf_device_info(pf_handle_t* h, char* name_out, size_t name_cap, int32_t* cu_count,
                   uint64_t* hbm_bytes);
/* PCI address of the handle's device (hipDeviceProp_t pciDomainID / pciBusID / pciDeviceID), so that a caller
 * can find the same GPU in tools that index in PCI order (rocm_smi) whatever HIP_VISIBLE_DEVICES says. */
int pf_device_pci(pf_handle_t* h, int32_t* domain, int32_t* bus, int32_t* device);

/* Single-GPU emulation of pf_forward_sharded over `nshards` ranks (test backend): same kernels
 * and per-shard workspaces as real ranks, the RCCL all-reduces replaced by device-side sums.
 * idx: host uint8 [B][N][L]; out: host float [B][P]. */
int pf_forward_shards_emulated(pf_handle_t* h, const uint8_t* idx, int32_t B, int32_t N, int32_t L,
                               int32_t nshards, float* out);

/* Hardware-layout self test: one wave exercises the cross-lane primitives and one
 * MFMA with known operands; `out` receives 2304 floats (layout in
 * phyloformer_amd/csrc/pf_device.hip.h::k_selftest).  tests/test_gpu_parity.py::test_hardware_layout_selftest
 * checks them against the layout the kernels assume. */
int pf_selftest(pf_handle_t* h, float* out);

/* ---- softmax multi-head attention ---------------------------------------
 *
 * The reference's MultiHeadAttention (phyloformer/attention.py:53-91: q/k/v projections :64-78,
 * QK^T / sqrt(head_dim) :81-82, softmax :83, PV :85, out_proj :89).  Nothing in the reference
 * instantiates it and no checkpoint fits it, so it is not part of pf_forward; it is provided as a
 * stand-alone operator with the module's call surface.  x, y: float [B][R][C][64]; attention runs
 * along C, independently for every (b, r) and each of the 4 heads.  Weights are nn.Linear tensors
 * ([out][in] row-major + bias).  The object shares its parent handle's device and stream and must be
 * destroyed before it.
 * Range (ABI 5): every contraction splits its operands into two fp16 limbs, so x, the weights and the
 * q / k / v projections must stay below 65504 in magnitude (any LayerNorm-ed activation does, by orders of
 * magnitude); beyond it the result is inf / NaN, not a wrong number. */
typedef struct pf_mha_weights_t {
    int32_t n_heads;     /* 4 */
    int32_t embed_dim;   /* 64 */
    const float *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo;
} pf_mha_weights_t;
typedef struct pf_mha pf_mha_t;
/* A handle with a device and a stream but no Phyloformer weights, for callers that only use the
 * stand-alone operators; pf_forward* on it fail with PF_ESTATE. */
int pf_create_bare(int device, pf_handle_t** out);
int pf_mha_create(pf_handle_t* h, const pf_mha_weights_t* w, pf_mha_t** out);
int pf_mha_destroy(pf_mha_t* m);
/* host buffers, synchronous */
int pf_mha_forward(pf_mha_t* m, const float* x, int32_t B, int32_t R, int32_t C, float* y);
/* device buffers, asynchronous on the parent handle's stream */
int pf_mha_forward_device(pf_mha_t* m, const float* d_x, int32_t B, int32_t R, int32_t C, float* d_y);

/* ---- host-side file formats of the CLI (no GPU, callable without a handle) --------------------
 *
 * pf_parse_fasta replaces load_alignment (phyloformer/data.py:11-31): `data[len]` is the whole file;
 * lines are split on '\n' and stripped of ASCII white space, a line starting with '>' opens a record
 * (id = rest of the line), other non-empty lines are residues of the current record.  Writes residue
 * indices uint8 [N][L] to `idx` (NULL = only measure) and, per record, the (offset, length) of its id
 * inside `data` to id_spans[2*N] (NULL = skip; at most max_seqs records).  Returns PF_OK and N, L, or
 *   PF_FASTA_EBYTE     byte outside the alphabet, *detail = the byte   (KeyError, data.py:26)
 *   PF_FASTA_ERAGGED   records of different lengths, *n_out still set  (ValueError from one_hot/stack)
 *   PF_FASTA_ENOHEADER residues before the first '>'                   (IndexError, data.py:26)
 *   PF_FASTA_EEMPTY    no record at all (RuntimeError from one_hot in the reference; so is N records of
 *                      length 0, which returns PF_OK with *l_out = 0)
 *   PF_FASTA_ECAP      idx_cap or max_seqs too small
 *   PF_FASTA_EUTF8     a header that bytes.decode("utf8") refuses, *detail = offset of the id in `data`
 *                      (UnicodeDecodeError at that line, data.py:22)
 * Errors are reported in file order: the first offending line decides, as in the reference's loop.
 */
#define PF_FASTA_EBYTE (-16)
#define PF_FASTA_ERAGGED (-17)
#define PF_FASTA_ENOHEADER (-18)
#define PF_FASTA_EEMPTY (-19)
#define PF_FASTA_ECAP (-20)
#define PF_FASTA_EUTF8 (-21)
int pf_parse_fasta(const char* data, int64_t len, uint8_t* idx, int64_t idx_cap, int64_t* id_spans,
                   int32_t max_seqs, int32_t* n_out, int32_t* l_out, int64_t* detail);

/* pf_format_phylip replaces vec_to_phylip (infer_alns.py:14-25): "N\n" then, per sequence,
 * "<id> d0 d1 ... dN-1\n" with "%.10f" entries of the symmetrised matrix (zero diagonal).
 * preds: float [N(N-1)/2], pairs (i<j) lexicographic; ids: N NUL-terminated strings.
 * Returns the text length in bytes (not NUL-terminated); nothing past `cap` is written, so a
 * first call with out = NULL, cap = 0 sizes the buffer. */
int64_t pf_format_phylip(const float* preds, int32_t n, const char* const* ids, char* out, int64_t cap);
/* Same with explicit id lengths (ids may then hold NUL bytes, as the reference's str ids may). */
int64_t pf_format_phylip_n(const float* preds, int32_t n, const char* const* ids, const int64_t* id_lens, char* out,
                           int64_t cap);

/* pf_nj_newick_n replaces skbio.tree.nj + the tree's text for the CLI's --trees (infer_alns.py:62-64,120-123):
 * neighbour joining (Saitou & Nei) on the symmetrised matrix of preds, float64 arithmetic, negative branch lengths
 * clamped to zero when clamp_negative != 0 (scikit-bio's default), the last three clusters joined at a trifurcation;
 * Newick text "(...);\n" with the ids as labels and Python-repr branch lengths - byte-identical to
 * phyloformer_amd/nj.py, which is pinned against FastME -m N trees of the reference's distances.  Sizing protocol as
 * pf_format_phylip (out = NULL, cap = 0 first). */
int64_t pf_nj_newick_n(const float* preds, int32_t n, const char* const* ids, const int64_t* id_lens, int32_t clamp_negative,
                       char* out, int64_t cap);

/* ---- many files per call, on native threads (ABI 4; tree_paths: ABI 5) ---------------------------
 *
 * The CLI loop (infer_alns.py:97-117) opens, parses, formats and writes one small file per alignment.
 * pf_fasta_batch_load reads and parses `count` files on up to `threads` native threads (pf_parse_fasta's
 * rules and status codes per file; PF_EIO with detail = errno when a file cannot be read) into a
 * library-owned batch object; pf_fasta_batch_infos fills per-file arrays of length count;
 * pf_fasta_batch_gather copies the residue indices of `count` (batch, file) entries, all of shape n x l,
 * into dst [count][n][l] - the input of pf_forward; pf_phylip_write_batch formats preds [count][n(n-1)/2]
 * as pf_format_phylip does, with the sequence ids the batch objects hold, and writes out_paths[k] - and, when
 * tree_paths is not NULL, the pf_nj_newick_n text of the same distances to tree_paths[k] - on up to
 * `threads` threads: status[k] = 0 or -errno.  A batch object is immutable after load: any number of threads
 * may read it; free it once, after the last use. */
typedef struct pf_fasta_batch pf_fasta_batch_t;
int pf_fasta_batch_load(const char* const* paths, int32_t count, int32_t threads, pf_fasta_batch_t** out);
void pf_fasta_batch_free(pf_fasta_batch_t* b);
int32_t pf_fasta_batch_count(const pf_fasta_batch_t* b);
int pf_fasta_batch_infos(const pf_fasta_batch_t* b, int32_t* status, int32_t* n, int32_t* l, int64_t* detail);
int pf_fasta_batch_id(const pf_fasta_batch_t* b, int32_t file, int32_t seq, const char** id, int64_t* len);
int pf_fasta_batch_gather(const pf_fasta_batch_t* const* batches, const int32_t* file_idx, int32_t count, int32_t n,
                          int32_t l, uint8_t* dst);
int pf_phylip_write_batch(const pf_fasta_batch_t* const* batches, const int32_t* file_idx, int32_t count, int32_t n,
                          const float* preds, const char* const* out_paths, const char* const* tree_paths, int32_t threads,
                          int32_t* status);

#ifdef __cplusplus
}
#endif
#endif /* PHYLOFORMER_AMD_H */
