// ref_pool.cpp -- TEST / BASELINE INFRASTRUCTURE, never linked into the product.
//
// A native std::thread pool that drives a CPU edlib implementation over a batch of units, one
// edlibAlign() per unit, exactly like the reference's own callers loop over it
// (/root/reference/apps/aligner/aligner.cpp:162-172; the function is re-entrant, SURVEY.md §8b "Threading").
// It is what bench.py's `cpu_baseline` leg times ("a std::thread pool of all host cores
// each calling edlibAlign on a disjoint slice") and what the full-batch parity checks of configs 4 and 5
// compare against.  The implementation is dlopen()ed by path:
//   oracle/_ref/libedlib_ref.so   the UNMODIFIED reference (symbols edlibAlign, edlibFreeAlignResult,
//                                 edlibAlignmentToCigar: /root/reference/edlib/include/edlib.h:146-271)
//   oracle/liboracle_edlib.so     the C99 restatement (oracle_align / oracle_free_result / oracle_cigar)
// Results come back as flat arrays (offsets + pools) so that Python compares a whole batch with numpy.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <malloc.h>
#include <string>
#include <thread>
#include <vector>

namespace {

// edlib.h:92-140, 162-218 (layout only; the header itself belongs to the reference)
struct EqPair { char first, second; };
struct AlignConfig { int k; int mode; int task; const EqPair* eqs; int neq; };
struct AlignResult {
    int status, editDistance;
    int* endLocations; int* startLocations; int numLocations;
    unsigned char* alignment; int alignmentLength, alphabetLength;
};
typedef AlignResult (*align_fn)(const char*, int, const char*, int, AlignConfig);
typedef void (*free_fn)(AlignResult);
typedef char* (*cigar_fn)(const unsigned char*, int, int);
typedef AlignResult (*oracle_align_fn)(const char*, int, const char*, int, int, int, int, const EqPair*, int);
typedef void (*oracle_free_fn)(AlignResult*);

struct Impl {
    void* h = nullptr;
    align_fn align = nullptr; free_fn release = nullptr; cigar_fn cigar = nullptr;
    oracle_align_fn oalign = nullptr; oracle_free_fn orelease = nullptr;
    bool load(const char* path, std::string& err) {
        h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!h) { err = dlerror(); return false; }
        align = (align_fn)dlsym(h, "edlibAlign");
        release = (free_fn)dlsym(h, "edlibFreeAlignResult");
        cigar = (cigar_fn)dlsym(h, "edlibAlignmentToCigar");
        if (!align) {
            oalign = (oracle_align_fn)dlsym(h, "oracle_align");
            orelease = (oracle_free_fn)dlsym(h, "oracle_free_result");
            cigar = (cigar_fn)dlsym(h, "oracle_cigar");
        }
        if (!(align && release) && !(oalign && orelease)) { err = "no edlibAlign / oracle_align in the library"; return false; }
        return true;
    }
    AlignResult run(const char* q, int qn, const char* t, int tn, const AlignConfig& c) const {
        return align ? align(q, qn, t, tn, c) : oalign(q, qn, t, tn, c.k, c.mode, c.task, c.eqs, c.neq);   // edlib_oracle.h: (k, mode, task)
    }
    void drop(AlignResult& r) const { if (release) release(r); else orelease(&r); }
};

struct Unit {
    int status = 0, ed = -1, nloc = 0, alen = 0, alpha = 0;
    bool hasEnds = false, hasStarts = false, hasAln = false;
    std::vector<int> ends, starts;
    std::vector<unsigned char> aln;
    std::string cigExt, cigStd;
};

}  // namespace

extern "C" {

struct RefPoolOut {
    int n, threads;
    double wall_seconds;          // the parallel region only
    int *status, *editDistance, *numLocations, *alphabetLength, *alignmentLength;
    unsigned char *hasEnds, *hasStarts, *hasAlignment;
    long long* locOff;            // [n+1] into ends / starts
    int *ends, *starts;
    long long* alnOff;            // [n+1] into alignment
    unsigned char* alignment;
    long long *cigExtOff, *cigStdOff;   // [n+1] each (only when wantCigar)
    char *cigExt, *cigStd;
    char error[256];
};

// Units i = 0..n-1: query = qpool[qoff[i]..qoff[i+1]), target = shared ? tpool[toff[0]..toff[1]) :
// tpool[toff[i]..toff[i+1]).  `select` (may be NULL) lists the units to run (nsel of them) -- the bounded
// sample of a large batch; outputs are indexed by position in `select`.  Work is handed out one unit at a
// time from an atomic counter (dynamic balance, as variable-length units need).  Returns NULL on failure
// to allocate; otherwise check out->error[0].
RefPoolOut* ref_pool_run(const char* libpath, int nthreads, const char* qpool, const long long* qoff,
                         const char* tpool, const long long* toff, int shared, int n,
                         const int* select, int nsel, int k, int mode, int task,
                         const char* eqPairs, int neq, int wantCigar)
{
    RefPoolOut* out = (RefPoolOut*)calloc(1, sizeof(RefPoolOut));
    if (!out) return nullptr;
    Impl impl; std::string err;
    if (!impl.load(libpath, err)) { snprintf(out->error, sizeof out->error, "%s", err.c_str()); return out; }
    const int cnt = select ? nsel : n;
    if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
    if (nthreads > cnt) nthreads = cnt > 0 ? cnt : 1;
    std::vector<EqPair> eqs((size_t)neq);
    for (int i = 0; i < neq; ++i) { eqs[i].first = eqPairs[2 * i]; eqs[i].second = eqPairs[2 * i + 1]; }
    AlignConfig cfg{k, mode, task, eqs.empty() ? nullptr : eqs.data(), neq};
    std::vector<Unit> res((size_t)cnt);
    std::atomic<int> next(0);
    auto work = [&]() {
        for (;;) {
            const int j = next.fetch_add(1, std::memory_order_relaxed);
            if (j >= cnt) break;
            const int u = select ? select[j] : j;
            const long long t0 = shared ? toff[0] : toff[u], t1 = shared ? toff[1] : toff[u + 1];
            AlignResult r = impl.run(qpool + qoff[u], (int)(qoff[u + 1] - qoff[u]), tpool + t0, (int)(t1 - t0), cfg);
            Unit& o = res[j];
            o.status = r.status; o.ed = r.editDistance; o.nloc = r.numLocations; o.alen = r.alignmentLength; o.alpha = r.alphabetLength;
            if (r.endLocations) { o.hasEnds = true; o.ends.assign(r.endLocations, r.endLocations + r.numLocations); }
            if (r.startLocations) { o.hasStarts = true; o.starts.assign(r.startLocations, r.startLocations + r.numLocations); }
            if (r.alignment) {
                o.hasAln = true; o.aln.assign(r.alignment, r.alignment + r.alignmentLength);
                if (wantCigar && impl.cigar) {
                    char* c = impl.cigar(r.alignment, r.alignmentLength, 1); if (c) { o.cigExt = c; free(c); }
                    c = impl.cigar(r.alignment, r.alignmentLength, 0); if (c) { o.cigStd = c; free(c); }
                }
            }
            impl.drop(r);
        }
    };
    // The reference allocates (and frees) its transformed copy of the target in every call (edlib.cpp:1425-1430:
    // 5 MB for config 2).  glibc serves a repeated request of exactly the size it last freed with a fresh mmap()
    // each time (the dynamic threshold is set to that chunk's size and the test is >=), so N threads spend their
    // time in page faults and TLB shootdowns instead of in edlibAlign.  A caller that cares pins the threshold;
    // so does this pool (REF_POOL_MALLOPT=0 leaves glibc alone) -- the reference itself is untouched.
    {
        const char* env = getenv("REF_POOL_MALLOPT");
        if (!(env && env[0] == '0')) { mallopt(M_MMAP_THRESHOLD, 32 << 20); mallopt(M_TRIM_THRESHOLD, 512 << 20); }
    }
    const auto w0 = std::chrono::steady_clock::now();
    {
        std::vector<std::thread> th;
        for (int i = 1; i < nthreads; ++i) th.emplace_back(work);
        work();
        for (auto& t : th) t.join();
    }
    out->wall_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
    out->n = cnt; out->threads = nthreads;
    // flatten
    auto ints = [&](size_t c) { return (int*)calloc(c ? c : 1, sizeof(int)); };
    auto bytes = [&](size_t c) { return (unsigned char*)calloc(c ? c : 1, 1); };
    auto offs = [&](size_t c) { return (long long*)calloc(c + 1, sizeof(long long)); };
    out->status = ints(cnt); out->editDistance = ints(cnt); out->numLocations = ints(cnt);
    out->alphabetLength = ints(cnt); out->alignmentLength = ints(cnt);
    out->hasEnds = bytes(cnt); out->hasStarts = bytes(cnt); out->hasAlignment = bytes(cnt);
    out->locOff = offs(cnt); out->alnOff = offs(cnt); out->cigExtOff = offs(cnt); out->cigStdOff = offs(cnt);
    for (int j = 0; j < cnt; ++j) {
        const Unit& o = res[j];
        out->status[j] = o.status; out->editDistance[j] = o.ed; out->numLocations[j] = o.nloc;
        out->alphabetLength[j] = o.alpha; out->alignmentLength[j] = o.alen;
        out->hasEnds[j] = o.hasEnds; out->hasStarts[j] = o.hasStarts; out->hasAlignment[j] = o.hasAln;
        out->locOff[j + 1] = out->locOff[j] + (long long)o.ends.size();
        out->alnOff[j + 1] = out->alnOff[j] + (long long)o.aln.size();
        out->cigExtOff[j + 1] = out->cigExtOff[j] + (long long)o.cigExt.size();
        out->cigStdOff[j + 1] = out->cigStdOff[j] + (long long)o.cigStd.size();
    }
    out->ends = ints((size_t)out->locOff[cnt]); out->starts = ints((size_t)out->locOff[cnt]);
    out->alignment = bytes((size_t)out->alnOff[cnt]);
    out->cigExt = (char*)bytes((size_t)out->cigExtOff[cnt]); out->cigStd = (char*)bytes((size_t)out->cigStdOff[cnt]);
    for (int j = 0; j < cnt; ++j) {
        const Unit& o = res[j];
        if (!o.ends.empty()) memcpy(out->ends + out->locOff[j], o.ends.data(), o.ends.size() * sizeof(int));
        for (size_t i = 0; i < o.ends.size(); ++i) out->starts[out->locOff[j] + i] = o.hasStarts ? o.starts[i] : -1;
        if (!o.aln.empty()) memcpy(out->alignment + out->alnOff[j], o.aln.data(), o.aln.size());
        if (!o.cigExt.empty()) memcpy(out->cigExt + out->cigExtOff[j], o.cigExt.data(), o.cigExt.size());
        if (!o.cigStd.empty()) memcpy(out->cigStd + out->cigStdOff[j], o.cigStd.data(), o.cigStd.size());
    }
    return out;
}

void ref_pool_free(RefPoolOut* o)
{
    if (!o) return;
    free(o->status); free(o->editDistance); free(o->numLocations); free(o->alphabetLength); free(o->alignmentLength);
    free(o->hasEnds); free(o->hasStarts); free(o->hasAlignment); free(o->locOff); free(o->ends); free(o->starts);
    free(o->alnOff); free(o->alignment); free(o->cigExtOff); free(o->cigStdOff); free(o->cigExt); free(o->cigStd);
    free(o);
}

// distinct (package, core) pairs of the online CPUs; 0 if the topology files are missing
int ref_pool_physical_cores(void)
{
    std::vector<long long> seen;
    for (int cpu = 0; cpu < 4096; ++cpu) {
        char p[160]; int core = -1, pkg = -1;
        snprintf(p, sizeof p, "/sys/devices/system/cpu/cpu%d/topology/core_id", cpu);
        FILE* f = fopen(p, "r"); if (!f) { if (cpu > 0) break; else continue; }
        if (fscanf(f, "%d", &core) != 1) core = -1;
        fclose(f);
        snprintf(p, sizeof p, "/sys/devices/system/cpu/cpu%d/topology/physical_package_id", cpu);
        f = fopen(p, "r"); if (f) { if (fscanf(f, "%d", &pkg) != 1) pkg = -1; fclose(f); }
        const long long key = ((long long)pkg << 32) | (unsigned)core;
        bool dup = false; for (long long s : seen) if (s == key) { dup = true; break; }
        if (!dup) seen.push_back(key);
    }
    return (int)seen.size();
}

}  // extern "C"
