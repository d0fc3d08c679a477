// api.hip -- the C ABI: the five symbols of reference edlib.h plus the additive
// batch surface of include/edlib_amd.h.  No C++ types cross this boundary.
#define EDLIB_SHARED
#define EDLIB_BUILD
#include "engine.hpp"

#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <system_error>
#include <atomic>
#include <thread>
#include <vector>

using namespace edlib_amd;

struct EdlibAmdBatch { Batch impl; };

static void fail_loudly(const char* where) {
    fprintf(stderr, "edlib (MI355X engine): %s failed: %s\n", where, last_error().c_str());
}

// No C++ exception may cross the C boundary (a caller written in C, or the reference's own clients, have no handler:
// std::bad_alloc out of a std::vector would be std::terminate).  Every entry point that can allocate runs inside this.
template <typename R, typename F>
static R guarded(const char* where, R failValue, F&& body) noexcept {
    try { return body(); }
    catch (const std::bad_alloc&) { try { set_error("%s: out of host memory", where); } catch (...) {} }
    catch (const std::exception& e) { try { set_error("%s: %s", where, e.what()); } catch (...) {} }
    catch (...) { try { set_error("%s: unknown C++ exception", where); } catch (...) {} }
    return failValue;
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
    const int rc = guarded("edlibAlign", 1, [&] {
        const int f = align_one_fused(query, queryLength, target, targetLength, config, &r);     // small pairs: one launch
        return f == 2 ? align_one(query, queryLength, target, targetLength, config, &r) : f;
    });
    if (rc) {
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
    return guarded("edlibAlignmentToCigar", static_cast<char*>(nullptr), [&]() -> char* {
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
    });
}

// ------------------------------------------------------------ edlib_amd.h

EDLIB_API int edlibAmdDeviceCount(void) { return device_count(); }
EDLIB_API const char* edlibAmdLastError(void) { return last_error().c_str(); }
EDLIB_API const char* edlibAmdVersion(void) { return "edlib-mi355x 0.1 (API of edlib 1.2.6)"; }

EDLIB_API EdlibAmdBatch* edlibAmdBatchCreateShared(const char* queries, const long long* queryOffsets,
                                                   int numQueries, const char* target, int targetLength,
                                                   EdlibAlignConfig config, int device) {
    EdlibAmdBatch* b = guarded("edlibAmdBatchCreateShared", static_cast<EdlibAmdBatch*>(nullptr), [] { return new EdlibAmdBatch; });
    if (!b) return nullptr;
    const long long toff[2] = {0, targetLength};
    const int rc = targetLength < 0 ? 1 : guarded("edlibAmdBatchCreateShared", 1, [&] {
        return b->impl.init(queries, queryOffsets, numQueries, target, toff, 1, config, device); });
    if (targetLength < 0) set_error("negative target length");
    if (rc) {
        delete b;
        return nullptr;
    }
    return b;
}

EDLIB_API EdlibAmdBatch* edlibAmdBatchCreatePairs(const char* queries, const long long* queryOffsets,
                                                  const char* targets, const long long* targetOffsets,
                                                  int numPairs, EdlibAlignConfig config, int device) {
    EdlibAmdBatch* b = guarded("edlibAmdBatchCreatePairs", static_cast<EdlibAmdBatch*>(nullptr), [] { return new EdlibAmdBatch; });
    if (!b) return nullptr;
    // a one-pair batch is also a shared-target batch
    if (guarded("edlibAmdBatchCreatePairs", 1, [&] {
            return b->impl.init(queries, queryOffsets, numPairs, targets, targetOffsets, numPairs, config, device); })) {
        delete b;
        return nullptr;
    }
    return b;
}

EDLIB_API int edlibAmdBatchRun(EdlibAmdBatch* b) {
    if (!b) { set_error("null batch"); return EDLIB_STATUS_ERROR; }
    return guarded("edlibAmdBatchRun", 1, [&] { return b->impl.run(); }) ? EDLIB_STATUS_ERROR : EDLIB_STATUS_OK;
}

EDLIB_API int edlibAmdBatchResults(EdlibAmdBatch* b, EdlibAlignResult* results) {
    if (!b || !results) { set_error("null argument"); return EDLIB_STATUS_ERROR; }
    return guarded("edlibAmdBatchResults", 1, [&] { return b->impl.results(results); }) ? EDLIB_STATUS_ERROR : EDLIB_STATUS_OK;
}

EDLIB_API int edlibAmdBatchResultsFlat(EdlibAmdBatch* b, int* status, int* editDistance, int* numLocations,
                                       int* alphabetLength, long long* locOffsets, int** endLocations,
                                       int** startLocations, long long* alnOffsets, unsigned char** alignment) {
    if (!b) { set_error("null argument"); return EDLIB_STATUS_ERROR; }
    return guarded("edlibAmdBatchResultsFlat", 1, [&] {
               return b->impl.resultsFlat(status, editDistance, numLocations, alphabetLength, locOffsets, endLocations,
                                          startLocations, alnOffsets, alignment); }) ? EDLIB_STATUS_ERROR : EDLIB_STATUS_OK;
}

EDLIB_API int edlibAmdBatchResultsView(EdlibAmdBatch* b, EdlibAmdResultsView* out) {
    if (!b || !out) { set_error("null argument"); return EDLIB_STATUS_ERROR; }
    return guarded("edlibAmdBatchResultsView", 1, [&] { return b->impl.resultsView(out); }) ? EDLIB_STATUS_ERROR : EDLIB_STATUS_OK;
}

EDLIB_API int edlibAmdBatchCigarView(EdlibAmdBatch* b, EdlibCigarFormat cigarFormat, const char** chars, const long long** offsets) {
    if (!b) { set_error("null argument"); return EDLIB_STATUS_ERROR; }
    return guarded("edlibAmdBatchCigarView", 1, [&] { return b->impl.cigarView((int)cigarFormat, chars, offsets); }) ? EDLIB_STATUS_ERROR : EDLIB_STATUS_OK;
}

EDLIB_API void edlibAmdFreeResults(EdlibAlignResult* results, int n) {
    if (!results) return;
    for (int i = 0; i < n; ++i) {
        edlibFreeAlignResult(results[i]);
        results[i].endLocations = nullptr; results[i].startLocations = nullptr; results[i].alignment = nullptr;
    }
}

EDLIB_API void edlibAmdTrim(void) { (void)guarded("edlibAmdTrim", 0, [] { pool_trim(); return 0; }); }

EDLIB_API int edlibAmdBatchStats(EdlibAmdBatch* b, EdlibAmdBatchStats* out) {
    if (!b || !out) { set_error("null argument"); return EDLIB_STATUS_ERROR; }
    (void)guarded("edlibAmdBatchStats", 0, [&] { b->impl.finishStats(); return 0; });
    *out = b->impl.stats;
    return EDLIB_STATUS_OK;
}

EDLIB_API void edlibAmdBatchDestroy(EdlibAmdBatch* b) { delete b; }

// The sequences of a one-shot call, packed back to back.  The pack goes straight into pinned staging (cached by the
// library) so that the upload that follows runs at link rate with no second host copy, and large packs are cut over
// a few host threads (1M x 150 bp = 150 MB of 150-byte memcpy: ~35 ms on one thread).
struct Packed {
    PinBuf pin; std::vector<char> small; std::vector<long long> off;
    const char* data() const { return pin.p ? reinterpret_cast<const char*>(pin.p) : small.data(); }
};

// false if a length is negative (edlibAlign answers EDLIB_STATUS_ERROR for those: so does the batch)
static bool pack(const char* const* seqs, const int* lens, int n, Packed& out) {
    out.off.assign(n + 1, 0);
    for (int i = 0; i < n; ++i) {
        if (lens[i] < 0) { set_error("negative sequence length at index %d", i); return false; }
        out.off[i + 1] = out.off[i] + lens[i];
    }
    const size_t bytes = (size_t)out.off[n] + 1;
    char* dst;
    if (bytes >= (1u << 20) && device_count() > 0 && out.pin.alloc(bytes) == hipSuccess) dst = reinterpret_cast<char*>(out.pin.p);
    else { (void)hipGetLastError(); out.pin.release(); out.small.resize(bytes); dst = out.small.data(); }
    auto copy = [&](int lo, int hi) { for (int i = lo; i < hi; ++i) if (lens[i] > 0) memcpy(dst + out.off[i], seqs[i], (size_t)lens[i]); };
    const int nthreads = bytes >= (32u << 20) ? host_threads(6) : 1;
    int done = 0;
    if (nthreads > 1) {
        std::vector<std::thread> th;
        th.reserve(nthreads);
        ThreadJoiner join(th);
        try {
            for (int t = 0; t < nthreads; ++t) {
                const int hi = (int)((long long)n * (t + 1) / nthreads);
                th.emplace_back(copy, done, hi);
                done = hi;
            }
        } catch (const std::system_error&) {}          // thread limit: this thread copies the rest
    }
    if (done < n) copy(done, n);
    return true;
}

// Devices the one-shot batch entry points shard over (contiguous slices of the units,
// target replicated, one host thread + one stream per device, no collective).  Default: device 0.
// EDLIB_AMD_DEVICES=all | "0,1,2,..." (a device may be listed twice: used by the tests on a 1-GPU box).
static std::vector<int> oneshot_devices() {
    std::vector<int> devs;
    const char* env = getenv("EDLIB_AMD_DEVICES");
    const int ndev = device_count();
    if (env && !strcmp(env, "all")) { for (int d = 0; d < ndev; ++d) devs.push_back(d); }
    else if (env && *env) {
        for (const char* p = env; *p;) {
            char* e; const long d = strtol(p, &e, 10);
            if (e == p) break;
            if (d >= 0 && d < ndev) devs.push_back((int)d);
            p = (*e == ',') ? e + 1 : e;
        }
    }
    if (devs.empty()) devs.push_back(default_device());
    return devs;
}

// Runs units [lo, hi) of a packed batch on one device.  toff == nullptr: shared target.
static int run_shard(const char* q, const long long* qoff, const char* t, const long long* toff, int targetLength,
                     int lo, int hi, EdlibAlignConfig config, int device, EdlibAlignResult* results, std::string* err) {
    EdlibAmdBatch* b = toff ? edlibAmdBatchCreatePairs(q, qoff + lo, t, toff + lo, hi - lo, config, device)
                            : edlibAmdBatchCreateShared(q, qoff + lo, hi - lo, t, targetLength, config, device);
    int rc = b ? edlibAmdBatchRun(b) : EDLIB_STATUS_ERROR;
    if (rc == EDLIB_STATUS_OK) rc = edlibAmdBatchResults(b, results + lo);
    if (rc != EDLIB_STATUS_OK) *err = last_error();
    if (b) edlibAmdBatchDestroy(b);
    return rc;
}

static int run_sharded(const char* q, const long long* qoff, const char* t, const long long* toff, int targetLength,
                       int n, EdlibAlignConfig config, EdlibAlignResult* results, const char* where) {
    const std::vector<int> devs = oneshot_devices();
    const int world = (int)std::min<size_t>(devs.size(), (size_t)std::max(1, n));
    std::vector<int> rc(world, EDLIB_STATUS_OK);
    std::vector<std::string> err(world);
    const int per = (n + world - 1) / world;                     // same rule as edlib_amd/parallel.py
    // Independent PAIRS vary in work (query x target cells): static slices leave devices idle behind the one that drew the
    // long pairs.  SURVEY.md 8e: a shared queue of chunks of about equal work (contiguous unit ranges, ~8 per device, cut
    // where the running sum of qlen * tlen crosses the next multiple of total / chunks) that the device threads pull from
    // with one atomic counter.  Shared-target read batches (equal reads) and EDLIB_AMD_SHARD=static keep the static slices.
    std::vector<int> cuts;                                       // chunk c = units [cuts[c], cuts[c + 1])
    {
        const char* how = getenv("EDLIB_AMD_SHARD");
        const bool queue = toff != nullptr && world > 1 && n >= 2 * world && !(how && !strcmp(how, "static"));
        if (queue) {
            double total = 0;
            for (int i = 0; i < n; ++i) total += (double)(qoff[i + 1] - qoff[i] + 1) * (double)(toff[i + 1] - toff[i] + 1);
            const int chunks = std::min(n, 8 * world);
            double acc = 0; int next = 1;
            cuts.push_back(0);
            for (int i = 0; i < n; ++i) {
                acc += (double)(qoff[i + 1] - qoff[i] + 1) * (double)(toff[i + 1] - toff[i] + 1);
                if (acc >= total * next / chunks && i + 1 < n) { cuts.push_back(i + 1); while (acc >= total * next / chunks) ++next; }
            }
            cuts.push_back(n);
        }
    }
    std::atomic<int> nextChunk{0};
    auto work = [&](int r) {                                     // a thread body: nothing may escape it
        rc[r] = guarded(where, (int)EDLIB_STATUS_ERROR, [&] {
            if (!cuts.empty()) {
                for (;;) {
                    const int c = nextChunk.fetch_add(1);
                    if (c + 1 >= (int)cuts.size()) return (int)EDLIB_STATUS_OK;
                    const int st = run_shard(q, qoff, t, toff, targetLength, cuts[c], cuts[c + 1], config, devs[r], results, &err[r]);
                    if (st != EDLIB_STATUS_OK) { nextChunk.store((int)cuts.size()); return st; }       // nobody starts another chunk
                }
            }
            const int lo = std::min(n, r * per), hi = std::min(n, lo + per);
            return (hi > lo || n == 0) ? run_shard(q, qoff, t, toff, targetLength, lo, hi, config, devs[r], results, &err[r])
                                       : (int)EDLIB_STATUS_OK;
        });
    };
    if (world == 1) work(0);
    else {
        std::vector<std::thread> th;
        th.reserve(world);
        int started = 0;
        {
            ThreadJoiner join(th);
            try { for (; started < world; ++started) th.emplace_back(work, started); }
            catch (const std::system_error&) {}        // thread limit: the shards without a thread run here, one by one
            for (int r = started; r < world; ++r) work(r);
        }
    }
    for (int r = 0; r < world; ++r) {
        if (rc[r] == EDLIB_STATUS_OK) continue;
        set_error("%s", err[r].c_str());
        fail_loudly(where);
        for (int i = 0; i < n; ++i) {                            // all or nothing: free what other shards produced
            if (results[i].status == EDLIB_STATUS_OK) edlibFreeAlignResult(results[i]);
            results[i] = blank_result(EDLIB_STATUS_ERROR);
        }
        return EDLIB_STATUS_ERROR;
    }
    return EDLIB_STATUS_OK;
}

EDLIB_API int edlibAlignBatchSharedTarget(const char* const* queries, const int* queryLengths, int numQueries,
                                          const char* target, int targetLength, EdlibAlignConfig config,
                                          EdlibAlignResult* results) {
    if (numQueries < 0 || targetLength < 0) { set_error("negative size"); return EDLIB_STATUS_ERROR; }
    for (int i = 0; i < numQueries; ++i) results[i] = blank_result(EDLIB_STATUS_ERROR);
    return guarded("edlibAlignBatchSharedTarget", (int)EDLIB_STATUS_ERROR, [&] {
        Packed q;
        if (!pack(queries, queryLengths, numQueries, q)) return (int)EDLIB_STATUS_ERROR;
        return run_sharded(q.data(), q.off.data(), target, nullptr, targetLength, numQueries, config, results,
                           "edlibAlignBatchSharedTarget");
    });
}

EDLIB_API int edlibAlignBatchPairs(const char* const* queries, const int* queryLengths,
                                   const char* const* targets, const int* targetLengths, int numPairs,
                                   EdlibAlignConfig config, EdlibAlignResult* results) {
    if (numPairs < 0) { set_error("negative size"); return EDLIB_STATUS_ERROR; }
    for (int i = 0; i < numPairs; ++i) results[i] = blank_result(EDLIB_STATUS_ERROR);
    return guarded("edlibAlignBatchPairs", (int)EDLIB_STATUS_ERROR, [&] {
        Packed q, t;
        if (!pack(queries, queryLengths, numPairs, q) || !pack(targets, targetLengths, numPairs, t)) return (int)EDLIB_STATUS_ERROR;
        return run_sharded(q.data(), q.off.data(), t.data(), t.off.data(), 0, numPairs, config, results,
                           "edlibAlignBatchPairs");
    });
}

}  // extern "C"
