/*
 * edlib_amd.h -- additive batch surface of the MI355X edit-distance engine.
 *
 * edlib.h (the reference's five symbols) stays untouched; everything here is
 * new.  The natural batching site in the reference is the per-query loop of
 * its CLI (apps/aligner/aligner.cpp:162-225: one edlibAlign() per query
 * against the same target) and, for pairwise work, any caller that loops over
 * edlibAlign() (bindings/python/edlib.pyx:128-129).  These entry points take
 * the whole loop at once so the GPU sees a batch.
 *
 * All functions are plain C ABI: pointers + sizes, no C++ or torch types.
 * Return value: EDLIB_STATUS_OK (0) / EDLIB_STATUS_ERROR (1) unless stated;
 * edlibAmdLastError() gives the reason.  There is no CPU fallback: without a
 * usable HIP device every compute entry point fails with EDLIB_STATUS_ERROR.
 */
#ifndef EDLIB_AMD_H
#define EDLIB_AMD_H

#include "edlib.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- process */

/* Number of usable HIP devices (0 when the runtime or a device is missing). */
EDLIB_API int edlibAmdDeviceCount(void);
/* Message of the last failing call on this thread ("" if none). */
EDLIB_API const char* edlibAmdLastError(void);
EDLIB_API const char* edlibAmdVersion(void);

/* ------------------------------------------------------- one-shot batches */

/* numQueries queries against ONE target: replaces the loop
 * `for q: results[q] = edlibAlign(queries[q], .., target, .., config)`
 * (apps/aligner/aligner.cpp:162-172).  results[] must hold numQueries
 * entries; each is exactly what edlibAlign() would have returned and is
 * released the same way (edlibFreeAlignResult / free).
 * Environment: EDLIB_AMD_DEVICES = "all" or a comma list of device ordinals
 * shards the units over several GPUs of the node (contiguous slices, target
 * replicated, one host thread + stream per device, no collective); default
 * is EDLIB_AMD_DEVICE if set, else the calling thread's current HIP device.  All-or-nothing: if any shard fails every result is ERROR. */
EDLIB_API int edlibAlignBatchSharedTarget(
    const char* const* queries, const int* queryLengths, int numQueries,
    const char* target, int targetLength,
    EdlibAlignConfig config, EdlibAlignResult* results);

/* numPairs independent (query, target) pairs: replaces
 * `for i: results[i] = edlibAlign(queries[i], .., targets[i], .., config)`. */
EDLIB_API int edlibAlignBatchPairs(
    const char* const* queries, const int* queryLengths,
    const char* const* targets, const int* targetLengths, int numPairs,
    EdlibAlignConfig config, EdlibAlignResult* results);

/* ------------------------------------------- resident batches (sessions) */
/* Upload once, run many times with everything resident in HBM; this is what
 * bench.py times.  Sequences are passed packed: `queries` is the
 * concatenation of all query bytes and queryOffsets[i]..queryOffsets[i+1]
 * delimits query i (numQueries+1 offsets). */

typedef struct EdlibAmdBatch EdlibAmdBatch;

EDLIB_API EdlibAmdBatch* edlibAmdBatchCreateShared(
    const char* queries, const long long* queryOffsets, int numQueries,
    const char* target, int targetLength,
    EdlibAlignConfig config, int device);

EDLIB_API EdlibAmdBatch* edlibAmdBatchCreatePairs(
    const char* queries, const long long* queryOffsets,
    const char* targets, const long long* targetOffsets, int numPairs,
    EdlibAlignConfig config, int device);

/* One pass of the device path over the resident batch (encode target, build
 * the query profiles, scan, merge; for LOC/PATH also start locations and
 * traceback), then wait for it.  Results stay on the device. */
EDLIB_API int edlibAmdBatchRun(EdlibAmdBatch* batch);

/* Copy the results of the last Run to the host as EdlibAlignResult[numQueries]
 * (malloc'd arrays, caller frees each with edlibFreeAlignResult).  On failure (EDLIB_STATUS_ERROR, e.g. host
 * memory exhausted) every entry is blank with status EDLIB_STATUS_ERROR: nothing is left for the caller to free.
 * No entry point of this library lets a C++ exception out. */
EDLIB_API int edlibAmdBatchResults(EdlibAmdBatch* batch, EdlibAlignResult* results);

/* The same results as flat arrays: no per-unit malloc, one array per field (what a numpy / columnar caller
 * wants).  status, editDistance, numLocations, alphabetLength: caller-provided int[numQueries];
 * locOffsets, alnOffsets: caller-provided long long[numQueries + 1]; *endLocations, *startLocations
 * (locOffsets[numQueries] ints; startLocations is NULL unless the task produced start locations) and
 * *alignment (alnOffsets[numQueries] op bytes) are malloc'd, the caller free()s them.  Any pointer may be NULL. */
EDLIB_API int edlibAmdBatchResultsFlat(EdlibAmdBatch* batch, int* status, int* editDistance, int* numLocations,
                                       int* alphabetLength, long long* locOffsets, int** endLocations,
                                       int** startLocations, long long* alnOffsets, unsigned char** alignment);

/* The same arrays WITHOUT copies: pointers into pinned host memory the batch owns, valid until the next Run / Destroy of
 * this batch.  For a batch of short pairs ("flat": every unit a pair of at most 16 blocks) the arrays are laid out by
 * device kernels -- the k filter, the -1 location of the padded last block, start locations, the dense op bytes, their
 * offsets by a prefix sum -- and arrive as one block at link rate; other batches build them once on the host.
 * startLocations is NULL unless the task produced start locations, alignment NULL unless it produced paths. */
typedef struct {
    int numUnits;
    const int* status;              /* [numUnits] EDLIB_STATUS_*                                  */
    const int* editDistance;        /* [numUnits] -1: no alignment within k                       */
    const int* numLocations;        /* [numUnits]                                                 */
    const int* alphabetLength;      /* [numUnits]                                                 */
    const long long* locOffsets;    /* [numUnits + 1] into endLocations / startLocations          */
    const int* endLocations;
    const int* startLocations;
    const long long* alnOffsets;    /* [numUnits + 1] into alignment (EDLIB_EDOP_* bytes)         */
    const unsigned char* alignment;
} EdlibAmdResultsView;
EDLIB_API int edlibAmdBatchResultsView(EdlibAmdBatch* batch, EdlibAmdResultsView* out);

/* edlibAlignmentToCigar() over every op string of the last Run (a TASK_PATH batch): *chars = the NUL-terminated CIGAR
 * strings one after the other, (*offsets)[i] = where string i starts ((*offsets)[numUnits] = all bytes); a unit without
 * an alignment has the empty string.  Same lifetime as the view.  Flat batches: made on the device. */
EDLIB_API int edlibAmdBatchCigarView(EdlibAmdBatch* batch, EdlibCigarFormat cigarFormat,
                                     const char** chars, const long long** offsets);

/* edlibFreeAlignResult() over results[0..n) (one call instead of n for binding languages). */
EDLIB_API void edlibAmdFreeResults(EdlibAlignResult* results, int n);

/* Releases the process-wide cache of device / pinned blocks and idle streams the library keeps between calls. */
EDLIB_API void edlibAmdTrim(void);

/* Counters of the last Run.  The struct GROWS AT ITS END between versions of this library (wide_retries came last) and
 * edlibAmdBatchStats() writes all of it: this additive surface is versioned with the library, not frozen like edlib.h --
 * build clients against the header of the library they load. */
typedef struct {
    double run_ms;          /* HIP-event time of the whole last Run on its stream           */
    double scan_ms;         /* HIP-event time of the dominant scan kernel(s) in that Run     */
    int scan_launches;      /* number of scan-kernel launches in that Run                    */
    long long cells;        /* sum over units of queryLength * targetLength (GCUPS numerator)*/
    long long word_steps;   /* 32-row word-column updates the scan kernels executed          */
    long long algo_bytes;   /* algorithmic bytes: target+query+Peq+results    */
    int path;               /* bit 0 reads-per-lane kernel, bit 1 block-per-lane kernel, bit 2 piece filter (long HW reads) */
    int overflow_units;     /* units whose end-location list needed the exact second pass    */
    int wide_retries;       /* launches of the many-wave kernel that gave up (not resident together / stalled) and were run again with one slot per unit */
} EdlibAmdBatchStats;

EDLIB_API int edlibAmdBatchStats(EdlibAmdBatch* batch, EdlibAmdBatchStats* out);
EDLIB_API void edlibAmdBatchDestroy(EdlibAmdBatch* batch);

#ifdef __cplusplus
}
#endif
#endif /* EDLIB_AMD_H */
