/*
 * acx.h -- C ABI of libacx.so, the MI355X (gfx950) all-pairwise cover-song
 * similarity engine behind acoss's CoverAlgorithm.similarity()/all_pairwise().
 *
 * The reference (furkanyesiler/acoss) is pure Python and has NO FFI of its own;
 * its only native boundary on this path is essentia's Python wrapper
 * (acoss/algorithms/rqa_serra09.py:60-67).  Each entry point below therefore
 * cites the reference INTERFACE it replaces; INTEGRATION.md shows the ctypes
 * binding a maintainer adds on the acoss side.
 *
 * Conventions
 *   - plain C types only; every host buffer is caller-allocated and
 *     caller-owned; the library owns device memory inside acx_ctx; no pointer
 *     outlives a call except the context.
 *   - every call returns ACX_OK (0) or a negative ACX_ERR_* code; the message
 *     is available from acx_last_error().  Nothing falls back to the CPU.
 *   - a context is single-owner (one host thread per context / per GPU).
 */
#ifndef ACX_H
#define ACX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 4): + acx_dev_alloc / _free / _read / _sync, acx_comm_id / _init / _destroy, acx_grid_allgather,
 *              acx_pair_grid_ranks; acx_ef_pool_end fails on tracks never handed over; uploads reject non-finite input
 *              (since the round-3 library, which still said 1).  The ctypes shim refuses a library of another version. */
/* 3 (round 5): + acx_device_info; acx_pair_grid_ranks carries a per-rank status through its exchanges (a failing rank fails the call
 *              on every rank instead of leaving the others in the all-gather); ACX_RCCL_LIB, when set, is the only librccl tried. */
#define ACX_ABI_VERSION 4

enum {
    ACX_OK = 0,
    ACX_ERR_INVALID = -1,      /* bad argument (NULL, index out of range, dim != expected)  */
    ACX_ERR_HIP = -2,          /* a HIP runtime call failed / no gfx950 device              */
    ACX_ERR_NOMEM = -3,        /* device scratch cannot hold even one pair                  */
    ACX_ERR_STATE = -4,        /* pool not uploaded                                         */
    ACX_ERR_SHORT = -5,        /* a track is shorter than the delay-embedding stack (essentia
                                  raises EssentiaException there)                           */
    ACX_ERR_UNSUPPORTED = -6   /* parameter combination not implemented on the device       */
};

typedef struct acx_ctx acx_ctx;

/* ---- context ------------------------------------------------------------ */

/* Create a context on HIP device `device`.  Returns NULL and sets *err on failure. */
acx_ctx *acx_create(int device, int *err);
void acx_destroy(acx_ctx *ctx);
/* Last error message of this context (ctx == NULL: of the last failed acx_create). */
const char *acx_last_error(const acx_ctx *ctx);
int acx_abi_version(void);
/* HIP_VERSION the library was compiled against and the version of the HIP runtime it is running on (0: unknown).
 * The ctypes shim records both (acoss_amd._lib.HIP_VERSIONS) and warns when the MAJOR versions differ; a process
 * that also holds PyTorch-ROCm runs on the runtime torch bundles (this image: built against 7.2, runs on 7.0 --
 * same major, recorded, no warning). */
int acx_hip_versions(int *build, int *runtime);
/* Which GPU is HIP device `device` of this process: PCI bus id ("0000:c1:00.0"), "name (gcn arch)", and the number of
 * devices the process can see (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES applied).  bench.py gathers one of these per
 * rank and refuses to report a multi-GPU figure when two ranks hold the same device.  No reference counterpart
 * (joblib workers share one host, algorithm_template.py:172-177). */
int acx_device_info(int device, char *pci_bus_id, int pci_len, char *name, int name_len, int *visible);
/* Upper bound (bytes) for the per-batch device scratch; 0 restores the default: env ACX_SCRATCH_GB, else 40 % of
 * device memory for the Serra09 batches and min(40 %, 36 GB) for the EarlyFusion arena (what one 128 x 128 grid tile
 * of 300-500-block tracks needs; larger arenas only cost allocation time). */
int acx_set_scratch_limit(acx_ctx *ctx, int64_t bytes);

/* ---- feature pool ------------------------------------------------------- */

/*
 * What an upload does with non-finite feature values.  Real feature files carry them -- the reference
 * zeroes NaN MFCCs itself (earlyfusion_traile.py:105) and hands everything else to essentia / numpy
 * unchecked, where one NaN frame spoils the scores of its own track.  Here it would spoil more: the
 * band kernel reads a neighbouring track's frames for the cells beyond a matrix's edge.  So every
 * acx_*upload* call scans what it receives, on the device:
 *   ACX_NONFINITE_REJECT (default)  the upload fails with ACX_ERR_INVALID, the message names the first
 *                                   offending track; no pool is left behind
 *   ACX_NONFINITE_ZERO              NaN / Inf values are replaced by 0 (what the reference does for MFCCs);
 *                                   acx_nonfinite_zeroed() tells how many the last upload replaced
 * NaN MFCC samples (acx_ef_block_features / acx_ef_upload_raw_pool) are zeroed under either policy.
 */
enum { ACX_NONFINITE_REJECT = 0, ACX_NONFINITE_ZERO = 1 };
int acx_set_nonfinite_policy(acx_ctx *ctx, int32_t policy);
int64_t acx_nonfinite_zeroed(const acx_ctx *ctx);

/*
 * Upload the packed feature pool: `frames` is row-major (sum_i T_i, dim) f32,
 * track i occupies rows offsets[i] .. offsets[i+1].  Replaces the per-track
 * feature cache of the reference (Serra09.load_features -> self.all_feats,
 * rqa_serra09.py:44-53): tracks are ALREADY pooled (x40 median) when uploaded.
 * dim must be 12 for the chroma algorithms.  Copies to HBM; the host buffers may
 * be released on return.
 */
int acx_upload_pool(acx_ctx *ctx, const float *frames, const int64_t *offsets,
                    int32_t n_tracks, int32_t dim);

/*
 * The same pool from RAW chroma: `raw` is row-major (sum_i T0_i, 12) f32, track i occupies rows
 * raw_offsets[i] .. raw_offsets[i+1]; the device takes the median over consecutive blocks of
 * `fac` frames (the last block of a track may be shorter) -- what Serra09.load_features does per
 * track with librosa.util.sync(chroma.T, arange(0, T0, fac), aggregate=np.median).T
 * (rqa_serra09.py:49-52; downsample_fac = 40 by default, <= 64 here).  Bit-identical to np.median
 * on f32 input.  pooled_offsets_out (n_tracks + 1 entries, may be NULL) receives the offsets of
 * the pooled tracks: ceil(T0_i / fac) frames each.
 */
int acx_upload_raw_pool(acx_ctx *ctx, const float *raw, const int64_t *raw_offsets, int32_t n_tracks,
                        int32_t dim, int32_t fac, int64_t *pooled_offsets_out);

/* Copy the f32 pool back (tests): `frames` has room for `capacity` floats. */
int acx_download_pool(acx_ctx *ctx, float *frames, int64_t capacity);

/* ---- Serra09 (OTI + delay embedding + CSM + mutual-kappa + Qmax) --------- */

/* Mirrors the constructor arguments of Serra09 (rqa_serra09.py:31-32) that reach
 * essentia (rqa_serra09.py:60-64), plus the switchable recalled-essentia details
 * documented in oracle/acx_oracle.c. */
typedef struct {
    int32_t m;           /* frameStackSize,   default 9 (device: 1..33) */
    int32_t tau;         /* frameStackStride, default 1 */
    float kappa;         /* binarizePercentile, default 0.095 */
    int32_t oti;         /* default 1 */
    float gamma_o;       /* disOnset, default 0.5 */
    float gamma_e;       /* disExtension, default 0.5 */
    int32_t embed_full;  /* 0: M = T - m*tau   1: M = T - (m-1)*tau */
    int32_t pct_mode;    /* 0 linear  1 essentia d0+d1  2 lower  3 nearest.  Position k = (n - 1) kappa; 0 and 1 both
                            interpolate d_(floor k) (ceil k - k) + d_(ceil k) (k - floor k) and differ only when k is an
                            exact integer: 1 evaluates the formula as recalled from essentia -- both weights are 0, the
                            threshold is 0 and the row stays empty (rows of n - 1 = 200, 400, ... cells at kappa =
                            0.095) --, 0 (the default, a DELIBERATE deviation until the pin kit says otherwise)
                            returns d_(k), the value the interpolation converges to.
                            tests/test_oracle_serra09.py::test_integer_percentile_position pins both. */
    int32_t oti_target;  /* 0 rotate reference  1 rotate query */
    int32_t dp_start;    /* 2 or 3 */
    int32_t inclusive;   /* 1: d <= eps  0: d < eps */
    int32_t dmax;        /* 0 Qmax ('serra09')  1 Dmax ('chen17') */
    int32_t arith;       /* ACX_ARITH_EXACT (default) or ACX_ARITH_F16X2, see below (ABI 2: new last field) */
} acx_serra09_params;

/*
 * Arithmetic of the frame Gram behind the distance matrix (everything else is the same code in both):
 *   ACX_ARITH_EXACT  v_mfma_f32_16x16x4_f32 chains: the f32 spec the oracle reproduces bit for bit
 *   ACX_ARITH_F16X2  opt-in, m = 9 only: every bin value as two fp16 terms x = h1 + h2 (22-23 significant bits; the matrix pipe
 *                    keeps fp16 subnormals), products x1 y1 + x1 y2 + x2 y1 on v_mfma_f32_16x16x32_f16 + 16x16x16_f16, which run
 *                    beside other waves' VALU work where an f32 MFMA blocks the SIMD.  As accurate as the f32 chain against f64
 *                    (2e-7 relative, scripts/ubench/mfma_f16_probe.hip) but NOT the same bits: scores are graded by north_star's
 *                    tolerance (|score difference| <= 2.0, identical MAP / MR on cover sets; tests/test_gpu_serra09.py), the
 *                    default and bench.py's headline stay exact.  Pairs beyond the band kernel (rows of more than 2041 cells)
 *                    take the exact streaming kernels in either mode.  INPUT RANGE: the split keeps its 22 bits only while the
 *                    first term is a normal, finite fp16 -- a pool whose largest magnitude lies outside [2^-8, 2^15] is
 *                    refused with ACX_ERR_UNSUPPORTED (HPCP / CREMA frames are normalised to a maximum of 1).
 */
enum { ACX_ARITH_EXACT = 0, ACX_ARITH_F16X2 = 1 };

void acx_serra09_default_params(acx_serra09_params *p);

/*
 * Scores for K (query, reference) track-index pairs: out[k] = max of the Qmax
 * matrix for pair (pairs[2k], pairs[2k+1]) -- exactly the scalar that
 * Serra09.similarity() stores into Ds[key][i][j] (rqa_serra09.py:55-69), for a
 * whole (K,2) idxs array in one call instead of one essentia round trip per pair.
 * pairs, out: host memory.  The pool must be uploaded.
 */
int acx_serra09_pairs(acx_ctx *ctx, const int32_t *pairs, int64_t K,
                      const acx_serra09_params *params, float *out);

/*
 * ChenFusion (latefusion_chen.py:58-72): for every pair ONE cross recurrence plot and two
 * alignments of it -- out[2k] = Qmax ('serra09'), out[2k+1] = Dmax ('chen17') -- the two
 * scalars ChenFusion.similarity() stores into Ds["qmax"][i][j] and Ds["dmax"][i][j].
 * params->dmax is ignored.  out: 2*K floats, host memory.
 */
int acx_chenfusion_pairs(acx_ctx *ctx, const int32_t *pairs, int64_t K,
                         const acx_serra09_params *params, float *out);

/*
 * One pair with intermediates, for parity tests (every pointer may be NULL):
 *   d2   (Mq*Mr) squared embedded distances (the ChromaCrossSimilarity distance
 *        matrix before sqrt, rqa_serra09.py:66)
 *   epsq (Mq), epsr (Mr)   the kappa-percentile thresholds on d = sqrt(d2)
 *   thrq (Mq), thrr (Mr)   the same thresholds moved to the d2 domain
 *                          (largest f32 x with sqrt(x) <=/< eps)
 *   oti, score, dims[2] = {Mq, Mr}
 */
int acx_serra09_debug_pair(acx_ctx *ctx, int32_t i, int32_t j,
                           const acx_serra09_params *params,
                           float *d2, float *epsq, float *epsr,
                           float *thrq, float *thrr,
                           int32_t *oti, float *score, int32_t *dims);

/*
 * The alignment alone (tests): Qmax ('serra09', params->dmax = 0) or Dmax ('chen17', dmax = 1) of a
 * given binary (M, N) uint8 cross recurrence plot -- the scalar CoverSongSimilarity returns for the
 * matrix ChromaCrossSimilarity produced (rqa_serra09.py:64,67; latefusion_chen.py:68-71).  Uses
 * params->gamma_o, gamma_e, dp_start, dmax; any M, N.  Non-{0,1} input -> ACX_ERR_INVALID.
 */
int acx_qmax_binary(acx_ctx *ctx, const uint8_t *R, int32_t M, int32_t N, const acx_serra09_params *params,
                    float *score);

/* Number of embedded frames for a pooled length T (0 if too short). */
int32_t acx_serra09_embed_len(int32_t T, const acx_serra09_params *params);

/* ---- SiMPle (similarity matrix profile) ---------------------------------- */

/*
 * Pool of SiMPle features in f64: `frames` is row-major (sum_i n_i, 12), TIME-major --
 * frame t of track i is the column t of what Simple.load_features(i) returns
 * (simple_silva.py:34-43: WIN/SKIP mean pooling + Hann smoothing + L2 column norm, done by
 * the host).  Independent of the f32 pool of acx_upload_pool.
 */
int acx_upload_pool_f64(acx_ctx *ctx, const double *frames, const int64_t *offsets,
                        int32_t n_tracks, int32_t dim);

/*
 * The same pool from RAW chroma (row-major (sum_i T0_i, 12) f32, C-contiguous per track as
 * deepdish loads it): Simple.load_features on the device for every track at once --
 * n_i = int(T0_i / skip) frames, frame k = mean of raw frames [k skip, k skip + win) clipped to
 * the track (f32, summed in time order like numpy reduces this axis; simple_silva.py:36-41), then
 * smooth(): hann(win_len_smooth + 2, sym) / sum, 'same' convolution along time, L2 norm per
 * frame (simple_silva.py:56-66), in f64.  Defaults of the reference: win 200, skip 100,
 * win_len_smooth 4.  pooled_offsets_out as in acx_upload_raw_pool.
 */
int acx_simple_upload_raw_pool(acx_ctx *ctx, const float *raw, const int64_t *raw_offsets, int32_t n_tracks,
                               int32_t dim, int32_t win, int32_t skip, int32_t win_len_smooth,
                               int64_t *pooled_offsets_out);

/* Copy the f64 pool back (tests): `frames` has room for `capacity` doubles. */
int acx_download_pool_f64(acx_ctx *ctx, double *frames, int64_t capacity);

/*
 * out[k] = -median(matrix profile) of the ORDERED pair (pairs[2k], pairs[2k+1]): OTI of the
 * second track toward the first, then simple_sim -- the value Simple.similarity() stores into
 * Ds['main'][i, j] (simple_silva.py:45-54, 68-126).  f64 like the reference.  sslen = SSLEN
 * (default 10, <= 16); tracks need sslen <= n_i <= 6000 pooled frames.  oti = 0 skips the
 * transposition (Simple.simple_sim alone, simple_silva.py:68).
 */
int acx_simple_pairs(acx_ctx *ctx, const int32_t *pairs, int64_t K, int32_t sslen, int32_t oti,
                     double *out);

/* ---- EarlyFusion (Tralie 2017) per-pair chain ---------------------------- */

/*
 * Block features of every track, as EarlyFusion.load_features() builds them
 * (earlyfusion_traile.py:100-154): for track i, blocks offsets[i] .. offsets[i+1];
 * mfccs (sum nb, dims[0]=650), ssms (sum nb, dims[1]=1225), chromas (sum nb, dims[2]=480,
 * 40 frames x 12 bins, bin fastest) f32 row-major, chroma_med (n_tracks, 12) f64.
 */
int acx_ef_upload_pool(acx_ctx *ctx, const float *mfccs, const float *ssms, const float *chromas,
                       const double *chroma_med, const int64_t *offsets, int32_t n_tracks,
                       const int32_t *dims);

/*
 * The same pool in slices of whole tracks, for collections whose block features the host cannot (or need
 * not) hold in one piece -- the reference keeps every track's blocks in a Python dict
 * (self.all_block_feats, earlyfusion_traile.py:100-154: 56 GB at DA-TACOS size).
 *   acx_ef_pool_begin   offsets (n_tracks + 1) and dims as in acx_ef_upload_pool; allocates
 *   acx_ef_pool_tracks  tracks [first_track, first_track + count): their rows of the three feature arrays
 *                       and their chroma medians, packed as in acx_ef_upload_pool.  The pointers may be
 *                       HOST or DEVICE memory (features that already live on the GPU, e.g. in a torch
 *                       tensor, are copied device to device).  STREAM CONTRACT: the copy is a blocking
 *                       hipMemcpy(hipMemcpyDefault) ordered against the NULL stream only -- a device source
 *                       written by kernels on another (non-blocking) stream must be COMPLETE before the call:
 *                       synchronise that stream (torch.cuda.synchronize() / hipStreamSynchronize) first.
 *                       Slices may arrive in any order and may be handed over again (the last copy wins)
 *   acx_ef_pool_end     checks that EVERY track was handed over (else ACX_ERR_STATE naming the first missing
 *                       track; the pool stays open, so the missing slices can still be supplied), then the
 *                       non-finite scan, row norms, the GEMM operands (two fp16 or three bf16 terms per value, acx_set_ef_gemm); the pool is usable from here on
 */
int acx_ef_pool_begin(acx_ctx *ctx, const int64_t *offsets, int32_t n_tracks, const int32_t *dims);
int acx_ef_pool_tracks(acx_ctx *ctx, int32_t first_track, int32_t count, const float *mfccs, const float *ssms,
                       const float *chromas, const double *chroma_med);
int acx_ef_pool_end(acx_ctx *ctx);

/* Block-feature parameters of EarlyFusion.load_features (ctor arguments blocksize,
 * mfccs_per_block, chromas_per_block; earlyfusion_traile.py:44-45): defaults 20, 50, 40. */
typedef struct {
    int32_t blocksize;          /* beats per block                                  */
    int32_t mfccs_per_block;    /* rows an MFCC block is resized to   (<= 64)        */
    int32_t chromas_per_block;  /* rows a chroma block is resized to  (<= 64)        */
} acx_ef_prep_params;

/*
 * EarlyFusion.load_features(i) for ONE track on the device (earlyfusion_traile.py:100-140, with
 * resize_block :214-247): from the track's chroma (n_chroma, 12) f32, MFCCs (n_mfcc, ncoef) f32
 * TIME-major (feats['mfcc_htk'].T; NaN -> 0 as at :108) and beat onsets (frame indices), the
 * n_blocks = n_beats - blocksize block features
 *   mfccs (n_blocks, mfccs_per_block * ncoef), ssms (n_blocks, mfccs_per_block (mfccs_per_block - 1) / 2),
 *   chromas (n_blocks, chromas_per_block * 12)  f32 row-major, chroma_med (12) f64
 * into host buffers (any may be NULL).  The anti-aliased resize restates skimage.transform.resize
 * (see acoss_amd/csrc/ef_prep_kernels.hpp; unpinned: skimage is absent from the reference's tree).
 */
int acx_ef_block_features(acx_ctx *ctx, const float *chroma, int64_t n_chroma, const float *mfcc, int64_t n_mfcc,
                          int32_t ncoef, const int64_t *onsets, int32_t n_beats, const acx_ef_prep_params *prep,
                          float *mfccs, float *ssms, float *chromas, double *chroma_med);

/*
 * The block-feature pool of a whole collection built on the device and kept there: the same as
 * acx_ef_block_features for every track followed by acx_ef_upload_pool, without the features
 * ever visiting the host.  chroma (sum n_chroma_i, 12), mfcc (sum n_mfcc_i, ncoef), onsets
 * (sum n_beats_i) packed, track i at *_offsets[i] .. *_offsets[i+1].  block_offsets_out
 * (n_tracks + 1, may be NULL) receives the block offsets.  Tracks with fewer than blocksize + 1
 * beats have no block.
 */
int acx_ef_upload_raw_pool(acx_ctx *ctx, const float *chroma, const int64_t *chroma_offsets, const float *mfcc,
                           const int64_t *mfcc_offsets, int32_t ncoef, const int64_t *onsets,
                           const int64_t *onset_offsets, int32_t n_tracks, const acx_ef_prep_params *prep,
                           int64_t *block_offsets_out);

typedef struct {
    double kappa;     /* csm_to_binary neighbourhood (EarlyFusion ctor kappa = 0.1)  */
    int32_t K;        /* getWCSM neighbours (ctor K = 10)                            */
} acx_ef_params;

/*
 * out[4k .. 4k+3] = Smith-Waterman scores (mfccs, ssms, chromas, early) of pair
 * (pairs[2k], pairs[2k+1]) -- what EarlyFusion.similarity() stores into
 * Ds['mfccs'|'ssms'|'chromas'|'early'][i, j] (earlyfusion_traile.py:157-198).
 */
int acx_earlyfusion_pairs(acx_ctx *ctx, const int32_t *pairs, int64_t K, const acx_ef_params *params,
                          float *out);

/*
 * Arithmetic of the three cross-similarity products (mfccs, ssms: get_csm, cross_recurrence.py:30-48; chromas:
 * get_csm_blocked_oti, :105-134 -- the reference's are BLAS sgemms in f32).  The matrix-pipe modes lay the pairs of a
 * batch out as dense rectangles (query tracks' blocks x reference tracks' blocks), so that the 256 x 128 tiles of the
 * GEMM are full and a track is read once per rectangle; chroma rows are kept bin-major, which turns the OTI roll into a
 * shift of the row by whole 16-byte pieces (blocks of a multiple of 8 frames; others: f32 MFMAs).
 *   ACX_EF_GEMM_F16X2 (default)   every value as TWO fp16 terms of x / s, s = the power of two that puts the largest
 *                                 |x| of the value's own ROW into [2^14, 2^15) (so a track's operands depend on the
 *                                 track alone); three fp16 MFMAs per cell and 32 k -- x1 y2, x2 y1, x1 y1, f32
 *                                 accumulation; x2 y2 (<= 2^-22 |x y|) stays below the accumulator's rounding and is
 *                                 not formed -- and the product rescaled by s_row s_column (exact).  An
 *                                 operand keeps 22 of its 24 significant bits: a relative rounding of 2^-23 per
 *                                 value, below the accumulation error of any f32 sgemm over K = 480 .. 1225.
 *   ACX_EF_GEMM_BF16X3            three-term bf16 splits (all 24 bits of every value), six bf16 MFMAs per cell and
 *                                 32 k: the dropped terms are below one f32 rounding of the product.  1.27 x the GEMM
 *                                 time of the default (round 3's default)
 *   ACX_EF_GEMM_F32               f32 MFMA (v_mfma_f32_16x16x4_f32: exact f32 products, f32 accumulation), one
 *                                 matrix at a time
 *   ACX_EF_GEMM_BF16X3_CHROMA_F32 the rectangles of BF16X3, chroma by f32 MFMAs (round 3's first kernel)
 * All meet the same bound against the f64 truth (tests/test_gpu_earlyfusion.py) and move as many scores against
 * f64-evaluated matrices as the reference's own f32 arithmetic does (tests/test_gpu_parity_sets.py, 124 750 pairs;
 * profiles/r04_parity_ef.json); scores that differ between them sit on a row-kappa threshold tie.  The split pool
 * holds ONE operand format (46 GB of fp16 terms or 69 GB of bf16 terms at DA-TACOS size, in one allocation sized for the larger):
 * changing between F16X2 and the bf16 modes re-splits the pool
 * on the next call.  mode -1 = the default.
 */
enum { ACX_EF_GEMM_BF16X3 = 0, ACX_EF_GEMM_F32 = 1,
       ACX_EF_GEMM_BF16X3_PAIRWISE = 2, /* mfccs / ssms in BF16X3's arithmetic, one matrix at a time (round 2's
                                           kernel; bit-identical matrices -- kept as the cross-check of the rectangle
                                           kernel), chroma by f32 MFMAs */
       ACX_EF_GEMM_BF16X3_CHROMA_F32 = 3, ACX_EF_GEMM_F16X2 = 4, ACX_EF_GEMM_DEFAULT = ACX_EF_GEMM_F16X2 };
int acx_set_ef_gemm(acx_ctx *ctx, int32_t mode);

/*
 * Arithmetic of getWCSM's kernel weights and of the fused matrix exp(-(W_mfcc + W_ssm + W_chroma)) (similarity_fusion.py:38-54,
 * earlyfusion_traile.py:176-183).  The fused matrix is only RANKED (csm_to_binary) afterwards.
 *   ACX_EF_FUSE_FAST (default)  exp(-C^2 / (2 (eps / 2)^2)) = exp2(-18 log2(e) (C / (r + c + C))^2): one v_rcp_f32, one v_exp_f32
 *                               per weight (about 1e-6 relative against the reference's numpy f32)
 *   ACX_EF_FUSE_EXACT           the reference's own operation order in f32 with IEEE divisions and expf (~12 x the instructions):
 *                               differs from numpy by the last bit of the exponentials only, so that a regression in the
 *                               selection / alignment kernels can be told from an approximation tie
 */
enum { ACX_EF_FUSE_FAST = 0, ACX_EF_FUSE_EXACT = 1 };
int acx_set_ef_fuse(acx_ctx *ctx, int32_t mode);

/* One pair with intermediates (tests): csm (3, M, N), fused (M, N), scores (4); any may be NULL. */
int acx_ef_debug_pair(acx_ctx *ctx, int32_t i, int32_t j, const acx_ef_params *params,
                      float *csm, float *fused, float *scores, int32_t *oti);

/* The same intermediates for pair `which` of a LIST of K pairs that runs as ONE batch (tests, ABI 4): the list goes through the
 * rectangle GEMM the way acx_earlyfusion_pairs sends it -- many pairs per workgroup tile, shared operands, sub-tiles dropped
 * into their own pairs' matrices -- so that a cell of a multi-pair rectangle can be held against an f64 product, not only the
 * one-pair rectangle of acx_ef_debug_pair.  scores: (K, 4), all pairs.  A list that needs more than one batch (65 535 pairs or
 * the scratch limit): ACX_ERR_UNSUPPORTED. */
int acx_ef_debug_pairs(acx_ctx *ctx, const int32_t *pairs, int64_t K, const acx_ef_params *params, int64_t which,
                       float *csm, float *fused, float *scores, int32_t *oti);

/* csm_to_binary(D, kappa) (cross_recurrence.py:136-161: exactly k = round(kappa N) cells per row;
 * ties at the k-th value are taken in column order) followed by smith_waterman_constrained, for
 * one (M, N) f32 matrix -- the tail of every feature's chain in earlyfusion_traile.py:165-197
 * (tests). */
int acx_csm_binary_sw(acx_ctx *ctx, const float *D, int32_t M, int32_t N, double kappa, float *score);

/* smith_waterman_constrained (alignment_tools.py:26-46) of one binary (M, N) uint8 matrix on
 * the device DP (tests against the reference goldens).  Non-{0,1} input -> ACX_ERR_INVALID
 * (the reference raises IOError). */
int acx_sw_binary(acx_ctx *ctx, const uint8_t *B, int32_t M, int32_t N, float *score);

/* ---- late fusion (N x N post-step) ---------------------------------------- */

/*
 * Similarity network fusion of m affinity matrices: the cross-diffusion loop of
 * doSimilarityFusionWs (similarity_fusion.py:146-186; EarlyFusion.do_late_fusion,
 * earlyfusion_traile.py:200-206; LateFusionChen.do_late_fusion) in f64 on the device.
 *   Ws[i]  (n, n) f64 row-major affinity matrix W_i (getW, :15-36 -- prepared by the host)
 *   Js[i]  (n, K) int32, Vs[i] (n, K) f64: the row-normalised K-nearest-neighbour kernel S_i of
 *          W_i (getS, :124-144) as K (column, weight) pairs per row
 *   out    (n, n) f64: the fused matrix, mean of the P_i after `niters` sweeps
 * P_i starts as the row-normalised W_i (getP, :101-122); per sweep and matrix
 * P_i <- S_i mean_{k != i}(P_k) S_i^T + reg_diag I, where from the second sweep on a matrix updated
 * earlier in the sweep is already seen by the later ones, as in the reference (:179).
 * 2 <= m <= 8.  Device memory: (2 m + 2) n^2 doubles.
 */
int acx_snf_fuse(acx_ctx *ctx, const double *const *Ws, const int32_t *const *Js, const double *const *Vs,
                 int32_t m, int32_t n, int32_t K, int32_t niters, double reg_diag, double *out);

/*
 * The whole of doSimilarityFusion (similarity_fusion.py:188-196) on the device: from m distance
 * matrices Ds[i] (n, n) f64 row-major -- getW (:15-36: symmetrise, zero diagonal, local scale =
 * mean of the K + 1 smallest of a row x (K + 1) / K, W = exp(-D^2 / (2 (mu eps)^2))), the kNN
 * kernels (getS, ties at the cut in column order), getP and the cross-diffusion loop as in
 * acx_snf_fuse.  out (n, n) f64: the fused matrix; Ws_out (may be NULL, entries may be NULL): the m
 * affinity matrices.  K <= 64.
 */
int acx_snf_fuse_dists(acx_ctx *ctx, const double *const *Ds, int32_t m, int32_t n, int32_t K, int32_t niters,
                       double reg_diag, double mu, double *out, double *const *Ws_out);

/* ---- the N x N pair grid -------------------------------------------------- */

/*
 * CoverAlgorithm.all_pairwise (algorithm_template.py:142-192) builds the list of all pairs
 * (itertools.combinations for symmetric algorithms, permutations otherwise, :168-171), cuts it into
 * 45 chunks for joblib (:172-177) and mirrors the result (D += D.T, :189-191).  Here the grid is cut
 * into tile x tile blocks of tracks (upper triangle incl. the diagonal blocks when symmetric),
 * each block costed by the sum of len_i * len_j over its pairs, and the blocks are dealt to `world`
 * ranks longest-processing-time-first.  A rank writes the scores of its blocks into one dense
 * buffer (block after block in deal order; a block is rows x cols x planes floats, planes
 * fastest); the only exchange of the whole path is one all-gather of those buffers.
 */
enum { ACX_ALGO_SERRA09 = 0, ACX_ALGO_CHENFUSION = 1, ACX_ALGO_SIMPLE = 2, ACX_ALGO_EARLYFUSION = 3 };

typedef struct {
    int32_t algo;       /* ACX_ALGO_*: 1 / 2 (qmax, dmax) / 1 / 4 (mfccs, ssms, chromas, early) score planes */
    int32_t symmetric;  /* 1: unordered pairs i < j   0: ordered pairs i != j (all_pairwise's `symmetric`) */
    int32_t tile;       /* tracks per block edge; 0: 128, halved while a rank would get fewer than 32 blocks */
    int32_t world;      /* number of ranks */
} acx_grid_spec;

typedef struct {
    int32_t row0, col0;   /* first track of the block's rows / columns */
    int32_t rows, cols;   /* tracks per side */
    int32_t rank;         /* owner */
    int32_t diagonal;     /* row0 == col0: only i < j (symmetric) or i != j is computed; the rest stays 0 */
    int64_t offset;       /* float offset of the block in its owner's buffer */
    double cost;          /* sum of len_i * len_j over the block's pairs */
} acx_grid_tile;

typedef struct {
    int32_t sslen;        /* SSLEN (Simple ctor, simple_silva.py:26-27), default 10 */
    int32_t oti;          /* 1: transpose the second track toward the first (Simple.oti) */
} acx_simple_params;

/*
 * The plan: pure host function of (track lengths, spec), identical on every rank.  `tiles` (may be
 * NULL) receives up to `capacity` blocks in deal order (cost descending), *n_tiles their number,
 * floats_per_rank[world] the size of every rank's score buffer, cost_per_rank[world] (may be NULL)
 * the dealt cost.  No device needed.
 */
int acx_grid_plan(const int64_t *lengths, int32_t n_tracks, const acx_grid_spec *spec, acx_grid_tile *tiles,
                  int64_t capacity, int64_t *n_tiles, int64_t *floats_per_rank, double *cost_per_rank);

/* Track lengths the grid of `algo` is planned on: pooled frames (Serra09 / ChenFusion: the f32
 * pool; SiMPle: the f64 pool) or blocks (EarlyFusion).  lengths: room for `capacity` entries. */
int acx_pool_lengths(acx_ctx *ctx, int32_t algo, int64_t *lengths, int32_t capacity, int32_t *n_tracks);

/*
 * Scores of blocks [first, first + count) of rank `rank`'s deal (count < 0: all of them) into
 * d_scores, a DEVICE buffer of floats_per_rank[rank] floats owned by the caller (e.g. the
 * storage of a torch tensor that is handed to the all-gather next).  The slice's part of the
 * buffer is zeroed first.  params: acx_serra09_params (Serra09, ChenFusion), acx_simple_params,
 * acx_ef_params.  Replaces the similarity(idxs) fan-out of algorithm_template.py:172-187.
 */
int acx_grid_run(acx_ctx *ctx, const acx_grid_spec *spec, const void *params, int32_t rank, int64_t first,
                 int64_t count, float *d_scores);

/*
 * Gathered rank buffers (HOST memory: `world` segments of rank_stride floats) -> the planes of
 * D: D[e][i * ld + j] = score plane e of pair (i, j), for the blocks [first, first + count) of
 * every rank; mirror != 0 also stores D[e][j * ld + i] (the D += D.T of
 * algorithm_template.py:189-191 on a matrix whose other triangle is zero).  Pure host function.
 */
int acx_grid_scatter(const int64_t *lengths, int32_t n_tracks, const acx_grid_spec *spec, const float *gathered,
                     int64_t rank_stride, int64_t first, int64_t count, float *const *D, int64_t ld,
                     int32_t mirror);

/* One GPU, whole grid: plan (world must be 1) + run + scatter into the caller's N x N host planes
 * (e.g. the float32 memmaps CoverAlgorithm.Ds holds). */
int acx_pair_grid(acx_ctx *ctx, const acx_grid_spec *spec, const void *params, float *const *D, int64_t ld,
                  int32_t mirror);

/* ---- multi-GPU inside the library: RCCL over xGMI, no Python ------------- */

/*
 * The reference fans its pair list out over joblib processes (algorithm_template.py:172-177); the Python host of this
 * library runs one process per GPU under torch.distributed (acoss_amd/dist.py).  A host WITHOUT Python gets the same path
 * from the library itself: one context (one GPU) per process or thread, one RCCL communicator rank per context.  librccl
 * is dlopen()ed on first use (env ACX_RCCL_LIB, a copy the process already holds, the one next to the HIP runtime, the
 * system's): single-GPU users never load it.
 *   acx_comm_id         rank 0: a fresh communicator id (ACX_COMM_ID_BYTES = sizeof(ncclUniqueId)); the HOST carries
 *                       it to the other ranks by whatever means it has (MPI, a socket, a file)
 *   acx_comm_init       every rank: join (collective: returns when all `world` ranks have called it)
 *   acx_grid_allgather  the one exchange of the path: every rank's score buffer (floats_per_rank floats at d_local,
 *                       what acx_grid_run filled) into d_gathered (world x floats_per_rank floats, DEVICE memory, rank r
 *                       at r * floats_per_rank) -- ncclAllGather on the library's stream, behind the kernels, device
 *                       to device over xGMI; returns when the buffer is complete
 *   acx_pair_grid_ranks the whole grid: plan, this rank's tiles, the all-gather, rank 0 scatters into its planes
 *                       (spec->world is ignored: the communicator's size is used; D may be NULL on ranks > 0).
 *                       A collective: every rank calls it with the same spec and params after uploading the same pool.
 *   acx_comm_destroy    leave (also done by acx_destroy)
 */
#define ACX_COMM_ID_BYTES 128
int acx_comm_id(void *id_out);
int acx_comm_init(acx_ctx *ctx, const void *id, int32_t rank, int32_t world);
int acx_comm_destroy(acx_ctx *ctx);
int acx_grid_allgather(acx_ctx *ctx, const float *d_local, float *d_gathered, int64_t floats_per_rank);
int acx_pair_grid_ranks(acx_ctx *ctx, const acx_grid_spec *spec, const void *params, float *const *D, int64_t ld,
                        int32_t mirror);

/* ---- device buffers for hosts without a GPU runtime of their own -------- */

/* acx_grid_run / acx_grid_allgather work on DEVICE buffers the caller owns.  A Python host takes them from torch; a host
 * that links no GPU runtime (and bench.py at one GPU, which then never imports torch) takes them from here: zero-filled
 * hipMalloc on the context's device; acx_dev_read drains the library's stream and copies to the host; acx_dev_sync =
 * hipDeviceSynchronize.  Buffers still alive are freed with the context. */
int acx_dev_alloc(acx_ctx *ctx, int64_t bytes, void **d_ptr);
int acx_dev_free(acx_ctx *ctx, void *d_ptr);
int acx_dev_read(acx_ctx *ctx, void *host_dst, const void *d_src, int64_t bytes);
int acx_dev_sync(acx_ctx *ctx);

/* ---- measurement -------------------------------------------------------- */

/* Per-kernel timing with HIP events recorded on the library's own stream around
 * every launch (bench.py's roofline leg).  Off by default.  While it is on, every
 * kernel runs on that one stream: the Serra09 alignment sweeps, which otherwise run
 * on a second stream beside the next band kernels, are timed alone, not beside
 * another launch -- a call is a few percent slower with the clocks on than without. */
int acx_profile_enable(acx_ctx *ctx, int on);
int acx_profile_reset(acx_ctx *ctx);
int acx_profile_count(const acx_ctx *ctx);
/* kernel `idx`: name (NUL-terminated into name[name_len]), accumulated
 * milliseconds, launches, and the number of pair-cells (Mq*Mr summed) it covered. */
int acx_profile_get(acx_ctx *ctx, int idx, char *name, int name_len,
                    double *ms, int64_t *launches, int64_t *cells);

/* device-side sqrt probe used by the parity tests (must be correctly rounded) */
int acx_debug_sqrt(acx_ctx *ctx, const float *in, int64_t n, float *out);
/* the sqrt of the EarlyFusion distance epilogues (v_sqrt_f32 + one Newton step: within 1 ulp, almost always exact) */
int acx_debug_ef_sqrt(acx_ctx *ctx, const float *in, int64_t n, float *out);

#ifdef __cplusplus
}
#endif
#endif /* ACX_H */
