// api.hip -- the C ABI: the five symbols of reference edlib.h plus the additive
// batch surface of include/edlib_amd.h.  No C++ types cross this boundary.
#define EDLIB_SHARED
#define EDLIB_BUILD
#include "engine.hpp"

#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace edlib_amd;

struct EdlibAmdBatch { Batch impl; };

static void fail_loudly(const char* where) {
    fprintf(stderr, "edlib (MI355X engine): %s failed: %s\n", where, last_error().c_str());
}

static EdlibAlignResult blank_result(int status) {
    EdlibAlignResult r;
    r.status = status; r.editDistance = -1;
    r.endLocations = nullptr; r.startLocations = nullptr; r.numLocations = 0;
    r.alignment = nullptr; r.alignmentLength = 0; r.alphabetLength = 0;
    return r;
}

extern "C" {

// ---------------------------------------------------------------- edlib.h

// reference edlib.cpp:1465-1475
EDLIB_API EdlibAlignConfig edlibNewAlignConfig(int k, EdlibAlignMode mode, EdlibAlignTask task,
                                               const EdlibEqualityPair* additionalEqualities,
                                               int additionalEqualitiesLength) {
    EdlibAlignConfig c;
    c.k = k; c.mode = mode; c.task = task;
    c.additionalEqualities = additionalEqualities;
    c.additionalEqualitiesLength = additionalEqualitiesLength;
    return c;
}

// reference edlib.cpp:1477-1479
EDLIB_API EdlibAlignConfig edlibDefaultAlignConfig(void) {
    return edlibNewAlignConfig(-1, EDLIB_MODE_NW, EDLIB_TASK_DISTANCE, nullptr, 0);
}

// reference edlib.cpp:1481-1485
EDLIB_API void edlibFreeAlignResult(EdlibAlignResult result) {
    free(result.endLocations);
    free(result.startLocations);
    free(result.alignment);
}

// reference edlib.cpp:146-301: a batch of one.  No CPU fallback: if the device
// path cannot run the result carries EDLIB_STATUS_ERROR and a line goes to stderr.
EDLIB_API EdlibAlignResult edlibAlign(const char* query, int queryLength, const char* target,
                                      int targetLength, const EdlibAlignConfig config) {
    EdlibAlignResult r = blank_result(EDLIB_STATUS_OK);
    if (queryLength < 0 || targetLength < 0) { r.status = EDLIB_STATUS_ERROR; return r; }
    if (align_one(query, queryLength, target, targetLength, config, &r)) {
        fail_loudly("edlibAlign");
        return blank_result(EDLIB_STATUS_ERROR);
    }
    return r;
}

// reference edlib.cpp:303-350: run-length encode EDLIB_EDOP_* codes.
EDLIB_API char* edlibAlignmentToCigar(const unsigned char* alignment, int alignmentLength,
                                      EdlibCigarFormat cigarFormat) {
    if (cigarFormat != EDLIB_CIGAR_EXTENDED && cigarFormat != EDLIB_CIGAR_STANDARD) return nullptr;
    static const char ext[4] = {'=', 'I', 'D', 'X'}, stdc[4] = {'M', 'I', 'D', 'M'};
    const char* letters = (cigarFormat == EDLIB_CIGAR_STANDARD) ? stdc : ext;
    std::string out;
    int i = 0;
    while (i < alignmentLength) {
        if (alignment[i] > 3) return nullptr;
        const char c = letters[alignment[i]];
        int run = 0;
        while (i < alignmentLength && alignment[i] <= 3 && letters[alignment[i]] == c) { ++run; ++i; }
        out += std::to_string(run);
        out += c;
    }
    char* s = static_cast<char*>(malloc(out.size() + 1));
    if (s) memcpy(s, out.c_str(), out.size() + 1);
    return s;
}

// ------------------------------------------------------------ edlib_amd.h

EDLIB_API int edlibAmdDeviceCount(void) { return device_count(); }
EDLIB_API const char* edlibAmdLastError(void) { return last_error().c_str(); }
EDLIB_API const char* edlibAmdVersion(void) { return "edlib-mi355x 0.1 (API of edlib 1.2.6)"; }

EDLIB_API EdlibAmdBatch* edlibAmdBatchCreateShared(const char* queries, const long long* queryOffsets,
                                                   int numQueries, const char* target, int targetLength,
                                                   EdlibAlignConfig config, int device) {
    EdlibAmdBatch* b = new (std::nothrow) EdlibAmdBatch;
    if (!b) { set_error("out of memory"); return nullptr; }
    const long long toff[2] = {0, targetLength};
    if (targetLength < 0 || b->impl.init(queries, queryOffsets, numQueries, target, toff, 1, config, device)) {
        delete b;
        return nullptr;
    }
    return b;
}

EDLIB_API EdlibAmdBatch* edlibAmdBatchCreatePairs(const char* queries, const long long* queryOffsets,
                                                  const char* targets, const long long* targetOffsets,
                                                  int numPairs, EdlibAlignConfig config, int device) {
    EdlibAmdBatch* b = new (std::nothrow) EdlibAmdBatch;
    if (!b) { set_error("out of memory"); return nullptr; }
    // a one-pair batch is also a shared-target batch
    if (b->impl.init(queries, queryOffsets, numPairs, targets, targetOffsets, numPairs, config, device)) {
        delete b;
        return nullptr;
    }
    return b;
}

EDLIB_API int edlibAmdBatchRun(EdlibAmdBatch* b) {
    if (!b) { set_error("null batch"); return EDLIB_STATUS_ERROR; }
    return b->impl.run() ? EDLIB_STATUS_ERROR : EDLIB_STATUS_OK;
}

EDLIB_API int edlibAmdBatchResults(EdlibAmdBatch* b, EdlibAlignResult* results) {
    if (!b || !results) { set_error("null argument"); return EDLIB_STATUS_ERROR; }
    return b->impl.results(results) ? EDLIB_STATUS_ERROR : EDLIB_STATUS_OK;
}

EDLIB_API int edlibAmdBatchStats(EdlibAmdBatch* b, EdlibAmdBatchStats* out) {
    if (!b || !out) { set_error("null argument"); return EDLIB_STATUS_ERROR; }
    *out = b->impl.stats;
    return EDLIB_STATUS_OK;
}

EDLIB_API void edlibAmdBatchDestroy(EdlibAmdBatch* b) { delete b; }

static int run_oneshot(EdlibAmdBatch* b, int n, EdlibAlignResult* results, const char* where) {
    if (!b) {
        fail_loudly(where);
        for (int i = 0; i < n; ++i) results[i] = blank_result(EDLIB_STATUS_ERROR);
        return EDLIB_STATUS_ERROR;
    }
    int rc = edlibAmdBatchRun(b);
    if (rc == EDLIB_STATUS_OK) rc = edlibAmdBatchResults(b, results);
    if (rc != EDLIB_STATUS_OK) {
        fail_loudly(where);
        for (int i = 0; i < n; ++i) results[i] = blank_result(EDLIB_STATUS_ERROR);
    }
    edlibAmdBatchDestroy(b);
    return rc;
}

static void pack(const char* const* seqs, const int* lens, int n, std::vector<char>& bytes,
                 std::vector<long long>& off) {
    off.assign(n + 1, 0);
    for (int i = 0; i < n; ++i) off[i + 1] = off[i] + (lens[i] > 0 ? lens[i] : 0);
    bytes.resize((size_t)off[n] + 1);
    for (int i = 0; i < n; ++i)
        if (lens[i] > 0) memcpy(bytes.data() + off[i], seqs[i], (size_t)lens[i]);
}

EDLIB_API int edlibAlignBatchSharedTarget(const char* const* queries, const int* queryLengths, int numQueries,
                                          const char* target, int targetLength, EdlibAlignConfig config,
                                          EdlibAlignResult* results) {
    std::vector<char> qb; std::vector<long long> qo;
    pack(queries, queryLengths, numQueries, qb, qo);
    return run_oneshot(edlibAmdBatchCreateShared(qb.data(), qo.data(), numQueries, target, targetLength, config, 0),
                       numQueries, results, "edlibAlignBatchSharedTarget");
}

EDLIB_API int edlibAlignBatchPairs(const char* const* queries, const int* queryLengths,
                                   const char* const* targets, const int* targetLengths, int numPairs,
                                   EdlibAlignConfig config, EdlibAlignResult* results) {
    std::vector<char> qb, tb; std::vector<long long> qo, to;
    pack(queries, queryLengths, numPairs, qb, qo);
    pack(targets, targetLengths, numPairs, tb, to);
    return run_oneshot(edlibAmdBatchCreatePairs(qb.data(), qo.data(), tb.data(), to.data(), numPairs, config, 0),
                       numPairs, results, "edlibAlignBatchPairs");
}

}  // extern "C"
